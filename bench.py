#!/usr/bin/env python3
"""bench.py — decode throughput of the MI355X hot path on BASELINE.json's configurations.

    python bench.py --gpus N --steps K --warmup W [--config 2|3|5]

Default (--config 2, the headline; with the sharding of config 4): a fixed batch of 64 independent
4K (3840x2160) VarDCT d1 frames (XYB, Gabor + EPF iters 2, XYB->sRGB), sharded by frame across the
N ranks (shard.frame_shard: 64 / N frames each, no data-path collective).  A "step" = every rank
renders its frames `--passes` (8) times — the full hot path V1-V8 + F1 + F2 + C1-C3 of SURVEY.md
§8(a) — from decoded state resident in HBM IN THE FORM THE ENTROPY DECODER EMITS IT (per-group
non-zero coefficient lists, JXLGPU_COEFF_GROUPED: every device pass a decoded frame needs is inside
the clock; there is no layout pass) to f32 RGB planes in HBM, through jxlgpu_vardct_render_batch
(one launch per stage for up to 32 frames).  Total work is fixed as N grows: "scaling": "strong".
`value` = 64 x passes x 8.2944 MP x K / wall time of the K timed steps (max over ranks).

Besides `value`, rank 0 reports in the same JSON line:
  roofline       HBM roofline of the dominant kernel group, HIP events on the library's own stream
                 inside the timed region (one bracket per batched launch);
  roofline_valu  the same launches against the VALU issue ceiling (the post stage is bound by f32
                 instruction issue under bit-exact arithmetic);
  verified       one frame per distinct workload downloaded after the timed region and compared
                 with the CPU oracle, bit for bit;
  gather_ms      the stitched-output step of config 4: u8 interleaved formatting on the device + ONE
                 gather of every rank's frames to rank 0 (RCCL over xGMI for N > 1), timed separately
                 from `value`; `value_with_gather` includes it;
  end_to_end     upload (grouped coefficient transport) + render + u8 download per frame, PCIe
                 inclusive (never `value`);
  cpu_baseline   the CPU oracle (C restatement of the reference's generic path, -O3 -march=native,
                 OpenMP inside a frame x frames in parallel) on this box's host cores, bounded sample.

--config 3: 8K Modular Squeeze lossy (i16) + XYB dequant + EPF + sRGB, frames rendered one by one.
--config 5: coded 4K VarDCT, EPF iters 3, 2x upsampling to 8K, Rec.2100 PQ (V1-V8 batched, post frame by frame;
            BASELINE's group sharding of one frame is covered by tests/test_shard.py).
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

W4K, H4K = 3840, 2160
HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
VALU_PEAK_GINSTR = 614.4    # 1024 SIMDs x 2.4 GHz / 4 cycles per wave64 instruction: what a packed / DPP / SGPR-operand instruction costs
VALU_PEAK_GINSTR_2CYC = 1228.8  # ... / 2 cycles: the micro-architecture guide's figure, and what plain v_fma / v_add / v_mul / v_and on
                            # VGPR operands reach with >= 4 waves per SIMD (profiles/r06_valu_cost_probe.txt: the post kernel's mix of
                            # 204 packed + 47 DPP + ~50 other 4-cycle and ~130 2-cycle instructions per row step costs ~3.5 cycles each)
# static VALU count of one row step of the streaming post kernel and its strip width (tools/isa_blocks.sh):
# packed kernel (two columns per lane, the default) / scalar kernel (JXLGPU_NO_PK)
POST_VALU_PER_ROW, POST_STRIP = (354, 56) if os.environ.get("JXLGPU_NO_PK") else (428, 120)
POST_HALO_ROWS = 8


def _owner_link_bound(gathered_gb, world, mp_per_step, gb_per_s_per_link=76.8):
    """Upper bound of a rooted (stitched on rank 0) job from the owner's inbound xGMI links: None at N = 1 or without a gather."""
    try:
        if not gathered_gb or world < 2:
            return None
        ms = gathered_gb / ((world - 1) * gb_per_s_per_link) * 1e3
        return {"GB_per_step": gathered_gb, "links": world - 1, "GB_per_s_per_link_inbound": gb_per_s_per_link,
                "ms_per_step": round(ms, 4), "value_bound": round(mp_per_step / (ms * 1e-3), 1), "unit": "MP/s"}
    except Exception:   # never let a reporting extra take the line down
        return None


def post_rows_per_seg(ih, iw, num_cus):
    """Rows per wave segment of the batched streaming post launch, as fused_prepare() (csrc/fused_kernels.hip) picks them:
    JXLGPU_BATCH_STREAM_ROWS if set, otherwise ONE resident round of waves (two per SIMD) per launch of 16 frames."""
    env = os.environ.get("JXLGPU_BATCH_STREAM_ROWS")
    if env and int(env) > 0:
        rows = int(env)
    else:
        fpl = int(os.environ.get("JXLGPU_BATCH_CHUNK", "0")) or (32 if os.environ.get("JXLGPU_NO_BATCH_OVERLAP") else 16)
        strips = -(-iw // 120)
        segs = max(1, (num_cus * 8 + fpl * strips // 2) // (fpl * strips))
        rows = max(32, -(-ih // segs))
    nseg = max(1, (ih + rows // 2) // rows)
    return (-(-ih // nseg) + 3) // 4 * 4


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--passes", type=int, default=None,
                    help="passes over the resident batch per step (default 8 for the headline config: a step is then "
                         "512 frame renders, ~60 ms — long enough for clocks to settle and for a utilisation sampler to see)")
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", type=int, default=2, choices=(2, 3, 5))
    ap.add_argument("--frames", type=int, default=None, help="frames in the whole job (default 64; 8 for configs 3 / 5)")
    ap.add_argument("--distinct", type=int, default=None, help="distinct synthetic frames in the job (block maps, coefficients, LF): "
                    "default 8 for the headline config, 2 for configs 3 / 5 (their oracle check renders 8K frames on the CPU)")
    ap.add_argument("--nz", default="0.15", help="VarDCT configs: fraction of the coefficients that are non-zero after quantisation "
                    "(SURVEY 8(d): 0.15); a comma list runs the whole measurement once per value, one JSON line each")
    ap.add_argument("--verify-frames", type=int, default=2, help="distinct frames checked against the oracle after the timed region")
    ap.add_argument("--transport", default="grouped", choices=("grouped", "dense_i32", "sparse_i16"),
                    help="coefficient transport of the resident input (VarDCT configs): the decoder's per-varblock "
                         "non-zero lists (default; consumed by the transform kernels directly) or dense planes")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=5.0, help="per CPU-baseline configuration")
    ap.add_argument("--no-verify", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip gather / end-to-end measurements")
    ap.add_argument("--gather", default="p2p", choices=("p2p", "rccl", "none"),
                    help="N > 1: how the stitched output reaches rank 0 inside the timed region — p2p: every rank's formatting kernels "
                         "store into rank 0's buffer through an IPC peer mapping (shard.PeerWriteGather; falls back to rccl if the "
                         "mapping cannot be made); rccl: overlapped dist.gather (shard.PipelinedGather); none: the output stays sharded")
    args = ap.parse_args()

    if args.distinct is None:
        args.distinct = 8 if args.config == 2 else 2
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world == 1 and args.gpus > 1:
        raise SystemExit("launch multi-GPU runs with torch.distributed.run (one rank per GPU)")

    import numpy as np
    import torch
    import torch.distributed as dist

    from jxl_oxide_amd import abi, runtime, shard

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP library is the thing measured (no CPU fallback)")
    if rank == 0:
        canary = runtime.gpu_canary(quiet=True)  # pure-HIP program first: a faulting box shows up here, by name
        print(f"CANARY {canary}", file=sys.stderr, flush=True)
    # test hook (one-GPU boxes): JXLGPU_BENCH_ONE_DEVICE=1 puts every rank on GPU 0 with gloo as the control backend, so
    # that the N > 1 code path (sharding, peer-write gather, verification) can be exercised without a second GPU
    one_device = bool(os.environ.get("JXLGPU_BENCH_ONE_DEVICE"))
    if one_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if one_device:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    ctx = runtime.Context(local_rank)

    def run_density(nz):
        n_total = args.frames or (64 if args.config == 2 else 8)
        job = make_job(args.config, args.distinct, args.transport, nz)
        # config 5 at N > 1 (BASELINE: "groups sharded across the GPUs"): every rank holds every frame and renders
        # its band of output rows of each (jxlgpu_vardct_render_region); everything else shards whole frames
        band_sharded = args.config == 5 and world > 1
        mine = list(range(n_total)) if band_sharded else list(shard.frame_shard(n_total, rank, world))
        wls = {d: job["make"](d) for d in sorted({i % args.distinct for i in mine})}  # untimed
        # a job may ask for frames to alternate between several contexts (config 3: JXLGPU_BENCH_CONTEXTS, default 3: each context
        # has its own streams, so consecutive frames overlap; the reference's caller pattern of one renderer per thread,
        # jxl-oxide-cli/src/decode.rs:293-304, driven from this one thread because every call is asynchronous)
        ctxs = [ctx] + [runtime.Context(local_rank) for _ in range(max(1, int(job.get("contexts", 1))) - 1)]
        frames = [job["upload"](ctxs[k % len(ctxs)], wls[i % args.distinct]) for k, i in enumerate(mine)]   # own device copy each; untimed
        mp_per_frame = job["out_w"] * job["out_h"] / 1e6

        passes = args.passes if args.passes else (8 if args.config == 2 else 1)

        # ---- N > 1: the stitched output is part of the job.  Every step's result is formatted on the device and
        # gathered to rank 0 (RCCL over xGMI) by shard.PipelinedGather: the gather of step k overlaps the kernels of
        # step k + 1, the host never waits inside the timed region, and `value` INCLUDES it.
        gather, gather_fmt, band, gather_mode = None, None, None, None
        if world > 1 and args.config in (2, 5):
            if band_sharded:
                bands = shard.band_rows(job["out_h"], world)
                band = bands[rank]
                hb = max(b[1] - b[0] for b in bands)
                gather_fmt = abi.FMT_U16   # "16-bit" output samples of config 5
                shape = (n_total, hb, job["out_w"], 3 * 2)   # as bytes: every RCCL build moves uint8
            else:
                gather_fmt = abi.FMT_U8
                shape = (-(-n_total // world), job["out_h"], job["out_w"], 3)
            slot_bytes = int(np.prod(shape[1:]))
            if args.gather == "p2p":
                try:
                    gather = shard.PeerWriteGather(ctx, slot_bytes, shape[0], dst=0)
                    gather_mode = "p2p"
                except Exception as e:  # noqa: BLE001  (every rank fails at the same call: the handle exchange is collective)
                    print(f"[bench] rank {rank}: peer-write gather unavailable ({type(e).__name__}: {e}); using the RCCL gather", file=sys.stderr)
                    gather = None
            if gather is None and args.gather != "none":
                gather = shard.PipelinedGather(shape, torch.uint8, "cuda", lib_stream=ctx.stream(), dst=0)
                gather_mode = "rccl"
        gstep = [0]

        def render_only():
            if band_sharded:
                for f in frames:
                    if band[1] > band[0]:   # more ranks than 8-row units: this rank has nothing to render
                        ctx.vardct_render_region(f, abi.STAGE_ALL, (0, band[0], job["out_w"], band[1] - band[0]), to_host=False)
            else:
                job["render"](ctx, frames)

        def step(with_gather=True):
            for _ in range(passes):
                render_only()
                if gather is not None and with_gather and frames:
                    if gather_mode == "p2p":
                        gather.write(frames, gather_fmt)
                    else:
                        buf = gather.slot(gstep[0])
                        shard.format_frames_into(ctx, frames, gather_fmt, buf)
                        gather.submit(gstep[0])
                    gstep[0] += 1

        def barrier():
            for c in ctxs:
                c.synchronize()
            torch.cuda.synchronize()
            if world > 1:
                dist.barrier()

        # ---- warmup; find the dominant kernel group with event brackets (one batch at a time)
        gather_error = None
        if gather is not None:
            # the first overlapped step runs under a guard: if this torch / RCCL build rejects any piece of the plumbing
            # (every rank fails at the same call), the job goes on without the gather instead of losing the measurement
            try:
                step()
                barrier()
            except Exception as e:  # noqa: BLE001
                gather_error = f"{type(e).__name__}: {e}"[:300]
                print(f"[bench] rank {rank}: overlapped gather unavailable ({gather_error}); timing the render only", file=sys.stderr)
                gather = None
        for _ in range(max(args.warmup, 1)):
            step()
        barrier()
        group_ms, group_n = {}, {}
        for g in job["groups"]:
            ctx.profile_select(g)
            step()
            ms, n = ctx.profile_read()
            group_ms[g], group_n[g] = ms, n
        dominant = max(group_ms, key=group_ms.get)
        ctx.profile_select(dominant)

        # ---- timed region: exactly K steps
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        barrier()
        elapsed = time.perf_counter() - t0
        prof_ms, prof_n = ctx.profile_read()
        ctx.profile_select(-1)
        if world > 1:
            t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if one_device else "cuda")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
        value = n_total * passes * args.steps * mp_per_frame / elapsed
        value_render_only = None
        if gather is not None:
            # the same K steps without the formatting + gather, for comparison (not the headline at N > 1)
            barrier()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                step(with_gather=False)
            barrier()
            e2 = time.perf_counter() - t0
            t = torch.tensor([e2], dtype=torch.float64, device="cpu" if one_device else "cuda")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            value_render_only = n_total * passes * args.steps * mp_per_frame / float(t.item())

        # ---- stitched output (config 4's gather): u8 formatting on the device + one gather, timed apart
        gather_ms = None
        if not args.no_extras and args.config == 2 and frames and world == 1:
            shard.gather_formatted_batch(ctx, frames, abi.FMT_U8)  # warm (allocations, RCCL channels)
            barrier()
            reps = 3
            t0 = time.perf_counter()
            for _ in range(reps):
                shard.gather_formatted_batch(ctx, frames, abi.FMT_U8)
            barrier()
            gather_s = (time.perf_counter() - t0) / reps
            if world > 1:
                t = torch.tensor([gather_s], dtype=torch.float64, device="cpu" if one_device else "cuda")
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                gather_s = float(t.item())
            gather_ms = gather_s * 1e3

        out = None
        if rank == 0:
            f0 = frames[0]
            # a batched step = ceil(frames / 32) launches of the group; prof_n brackets in K steps
            frames_per_launch = len(frames) * passes * args.steps / max(prof_n, 1) if job["batched"] else 1
            alg_frame = job["alg_bytes"](f0, dominant)
            avg_ms = prof_ms / max(prof_n, 1)
            achieved = alg_frame * frames_per_launch / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
            traffic, traffic_src = job["traffic"](dominant, frames_per_launch)
            roofline = {
                "bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_src,
                "kernel": job["group_names"][dominant], "avg_launch_ms": round(avg_ms, 4), "launches": int(prof_n),
                "frames_per_launch": round(frames_per_launch, 2),
                "algorithmic_bytes_per_launch": int(alg_frame * frames_per_launch),
                "group_ms_per_frame": {job["group_names"][g].split(":")[0]: round(group_ms[g] / max(len(frames) * passes, 1), 4) for g in group_ms},
                "pipeline_algorithmic_frac": round(job["alg_bytes"](f0, None) * n_total * passes * args.steps / elapsed / 1e9 / world / HBM_PEAK_GBS, 4),
                "traffic_ratio": None if not traffic else round(traffic / (alg_frame * frames_per_launch), 3),
                "concurrency": ("the batched launches of this group run while the V1-V8 launches of the next 16-frame chunk occupy part of "
                                "the CUs (separate streams): avg_launch_ms is the duration under that sharing; roofline_isolated has the "
                                "kernel alone") if (job["batched"] and not os.environ.get("JXLGPU_NO_BATCH_OVERLAP")) else None,
            }
            # What the job HAS to move with the resident input it actually reads: with the list transport the
            # 12 B/px of coefficient planes that SURVEY 8(d)'s 24.33 B/px counts are never read — the compulsory bytes are the
            # lists (4 B per non-zero coefficient + 6 B of counts per varblock), the side data and the 12 B/px written.
            if job.get("compulsory_bytes"):
                cb = job["compulsory_bytes"](wls[next(iter(wls))], f0)
                roofline["compulsory_bytes_per_frame"] = int(cb)
                roofline["pipeline_compulsory_frac"] = round(cb * n_total * passes * args.steps / elapsed / 1e9 / world / HBM_PEAK_GBS, 4)
                roofline["compulsory_note"] = ("list transport: non-zero lists + counts + side data read, f32 planes written; "
                                               "pipeline_algorithmic_frac credits SURVEY 8(d)'s 12 B/px of coefficient planes that this path never reads")
            # The timed region runs V1-V8 of chunk k+1 beside the post launch of chunk k (two streams): the bracket above is
            # the kernel's duration WHILE IT SHARES THE CUs.  The same kernel alone (a second context without the overlap,
            # 32 frames per launch as in rounds 2-3), bracketed the same way, for comparison with earlier rounds:
            roofline_isolated = None
            if args.config == 2 and job["batched"] and not args.no_extras:
                os.environ["JXLGPU_NO_BATCH_OVERLAP"] = "1"
                ctx2 = runtime.Context(local_rank)
                del os.environ["JXLGPU_NO_BATCH_OVERLAP"]
                fr2 = [job["upload"](ctx2, wls[mine[i % len(mine)] % args.distinct]) for i in range(32)]
                for _ in range(2):
                    job["render"](ctx2, fr2)
                ctx2.synchronize()
                ctx2.profile_select(dominant)
                for _ in range(6):
                    job["render"](ctx2, fr2)
                ms2, n2 = ctx2.profile_read()
                ctx2.profile_select(-1)
                for f in fr2:
                    f.free()
                ctx2.close()
                if n2 and ms2 > 0:
                    a2 = alg_frame * 32 / (ms2 / n2 * 1e-3) / 1e9
                    roofline_isolated = {"achieved": round(a2, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(a2 / HBM_PEAK_GBS, 4),
                                         "avg_launch_ms": round(ms2 / n2, 4), "frames_per_launch": 32,
                                         "what": "the dominant kernel group with JXLGPU_NO_BATCH_OVERLAP=1 (one stream, stage after stage): "
                                                 "nothing else on the CUs while it runs; NOT how the timed region ran"}
            roofline_valu = None
            if args.config == 2 and dominant == 2:
                # the streaming kernel's region and segmentation, as fused_prepare() lays them out
                # (JXLGPU_PK_TB=1: the packed kernel takes the top / bottom image rows itself — every row of the frame)
                tb = os.environ.get("JXLGPU_PK_TB", "0") not in ("", "0") and not os.environ.get("JXLGPU_NO_PK")
                iw, ih = (W4K - 4) // 8 * 8 - 16, (H4K if tb else (H4K - 4) // 8 * 8 - 16)
                rows = post_rows_per_seg(ih, iw, torch.cuda.get_device_properties(local_rank).multi_processor_count)
                segs = -(-ih // rows)
                strips = -(-iw // POST_STRIP)
                winstr = strips * segs * (rows + POST_HALO_ROWS) * POST_VALU_PER_ROW * frames_per_launch
                ach = winstr / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
                roofline_valu = {"bound": "valu", "achieved": round(ach, 1), "peak": VALU_PEAK_GINSTR, "unit": "G wave-instr/s",
                                 "frac": round(ach / VALU_PEAK_GINSTR, 4),
                                 "peak_2_cycles": VALU_PEAK_GINSTR_2CYC, "frac_at_2_cycles": round(ach / VALU_PEAK_GINSTR_2CYC, 4),
                                 "peaks": "`peak` = one wave64 instruction per 4 SIMD cycles (packed f32, DPP, SGPR / 3-operand integer forms: what "
                                          "profiles/r06_valu_cost_probe.txt measures for them); `peak_2_cycles` = the guide's v_fma_f32 figure, reached "
                                          "only by plain VGPR-operand f32 / logic instructions at >= 4 waves per SIMD; this kernel's mix averages ~3.5",
                                 "note": "streaming post kernel, f32 in the reference's operation order (no FMA contraction, "
                                         "correctly rounded division); a packed v_pk_*_f32 instruction counts once; halo rows "
                                         "and columns are recomputed (%d/%d x %d/%d)" % (
                                             POST_STRIP + 8, POST_STRIP, rows + POST_HALO_ROWS, rows),
                                 "rows_per_segment": rows, "valu_per_row_step": POST_VALU_PER_ROW}
            verified = None
            if not args.no_verify and not band_sharded:
                vw = dict(list(wls.items())[:max(1, args.verify_frames)])
                verified = job["verify"](ctx, frames, mine, vw, args.distinct)
            if not args.no_verify and gather is not None:
                # the stitched output as it arrived on rank 0: one frame that another rank rendered (config 2 / 4), or frame 0
                # reassembled from every rank's band (config 5), against the oracle's formatted render
                from oracle import pyoracle
                if gather_mode == "p2p":
                    # (every rank's stream has drained and a barrier has passed since the last write: barrier() above)
                    raw = gather.result()   # (world, slots, slot_bytes) uint8 on rank 0

                    class _Slot:   # the indexing the checks below use on the RCCL gather's tensors
                        def __init__(self, a):
                            self.a = a

                        def __getitem__(self, k):
                            return _Slot(self.a[k])

                        def cpu(self):
                            return self

                        def numpy(self):
                            return self.a
                    got_all = [_Slot(raw[r].reshape((shape[0],) + tuple(shape[1:]))) for r in range(world)]
                else:
                    got_all = gather.finish(gstep[0] - 1)
                if band_sharded:
                    wl = wls[0]
                    exp, _ = pyoracle.vardct_render(wl.desc(), abi.STAGE_ALL, job["out_w"], job["out_h"])
                    exp16 = pyoracle.format_output(exp, gather_fmt, 1)
                    ok = True
                    for r, (y0, y1) in enumerate(shard.band_rows(job["out_h"], world)):
                        ok &= bool(np.array_equal(got_all[r][0, :y1 - y0].cpu().numpy().view(np.uint16), exp16[y0:y1]))
                    gv = {"ok": ok, "what": "frame 0 reassembled on rank 0 from the %d gathered u16 bands == oracle render, formatted" % world}
                else:
                    other = list(shard.frame_shard(n_total, world - 1, world))[0]
                    wl = wls.get(other % args.distinct) or job["make"](other % args.distinct)
                    exp, _ = pyoracle.vardct_render(wl.desc(), abi.STAGE_ALL, job["out_w"], job["out_h"])
                    ok = bool(np.array_equal(got_all[world - 1][0].cpu().numpy(), pyoracle.format_output(exp, gather_fmt, 1)))
                    gv = {"ok": ok, "what": "first frame of rank %d as gathered on rank 0 (u8) == oracle render, formatted" % (world - 1)}
                verified = dict(verified or {"ok": True}, gathered=gv)
                verified["ok"] = bool(verified["ok"] and gv["ok"])
            e2e = None
            if not args.no_extras and args.config == 2:
                e2e = end_to_end(ctx, wls[mine[0] % args.distinct], mp_per_frame)
            cpu = None
            if not args.no_cpu_baseline and world == 1:  # reported at N = 1 only (bounded CPU sample)
                cpu = cpu_baseline(args.config, args.cpu_seconds)
            others = None
            if not args.no_extras and args.config == 2 and world == 1:
                others = other_configs()
            gathered_gb = None if gather is None else round(
                (gather.slot_bytes * gather.slots * (world - 1) if gather_mode == "p2p" else gather.bytes_to_dst / max(gstep[0], 1)) * passes / 1e9, 3)
            out = {
                "metric": job["metric"],
                "value": round(value, 1),
                "unit": "MP/s",
                "n_gpus": world,
                "steps": args.steps,
                "warmup": args.warmup,
                "ms_per_step": round(elapsed / args.steps * 1e3, 4),
                "higher_is_better": True,
                "scaling": "strong",
                "value_includes": (("render + device formatting whose stores land in rank 0's HBM through IPC peer mappings over xGMI (shard.PeerWriteGather: "
                                    "no collective, every rank over its own link)" if gather_mode == "p2p" else
                                    "render + device formatting + gather of the stitched output to rank 0 (overlapped: shard.PipelinedGather)")
                                   if gather is not None else "render (N = 1: the output is on the one GPU; formatting timed apart as gather_ms)"),
                "value_render_only": None if value_render_only is None else round(value_render_only, 1),
                "gather_error": gather_error,
                "gather_mode": gather_mode,
                "gathered_GB_per_step": gathered_gb,
                # what the owner's xGMI links allow at most for a ROOTED output: every other rank's bytes enter rank 0
                # over that rank's own link (p2p) — (world - 1) links x 76.8 GB/s inbound; `value` can approach, never pass, `value_bound`;
                # `value_render_only` (the output left sharded) is not subject to it.  Arithmetic only, nothing measured.
                "owner_link_bound": _owner_link_bound(gathered_gb, world, n_total * passes * mp_per_frame),
                "vs_baseline": None,
                "dtype": job["dtype"],
                "data": "synthetic",
                "config": {
                    "workload": job["workload"],
                    "frames_in_job": n_total,
                    "passes_per_step": passes,
                    "frames_per_gpu_per_step": len(frames) * passes,
                    "distinct_frames": args.distinct,
                "nz_fraction": nz if args.config != 3 else None,
                    "sharding": ("one frame = N bands of output rows, one per rank (shard.band_rows + jxlgpu_vardct_render_region), u16 bands gathered to rank 0"
                                 if band_sharded else
                                 "frames across ranks (shard.frame_shard), no data-path collective; the u8 output gathered to rank 0"),
                    "input": job.get("input", "decoded state resident in HBM"),
                    "launches": "jxlgpu_vardct_render_batch: one launch per stage for <= 32 frames" if job["batched"] else "one frame at a time",
                    "contexts": len(ctxs),
                },
                "roofline": roofline,
                "roofline_isolated": roofline_isolated,
                "roofline_valu": roofline_valu,
                "verified": verified,
                "gather_ms": None if gather_ms is None else round(gather_ms, 3),
                "value_with_gather": None if gather_ms is None else round(
                    n_total * passes * mp_per_frame / (elapsed / args.steps + passes * gather_ms * 1e-3), 1),
                "end_to_end": e2e,
                "other_configs": others,
                "cpu_baseline": cpu,
            }
        if gather is not None and gather_mode == "p2p":
            gather.close()   # collective: unmaps on the writers, frees on rank 0
        for f in frames:
            f.free()
        for c in ctxs:
            c.synchronize()
        for c in ctxs[1:]:
            c.close()
        if out is not None:
            print(json.dumps(out), flush=True)

    # one JSON line per density (default: the one density SURVEY §8(d) specifies)
    for nz in [float(v) for v in str(args.nz).split(",") if v.strip()]:
        run_density(nz)
    ctx.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


# ------------------------------------------------------------------------------------------------
def make_job(config, distinct, transport="grouped", nz=0.15):
    import numpy as np
    from jxl_oxide_amd import abi

    def pmc_traffic(pattern_names):
        def fn(dominant, frames_per_launch):
            src = next((os.path.join("profiles", n) for n in ("r06_pmc_hbm_traffic.json", "r05_pmc_hbm_traffic.json", "r04_pmc_hbm_traffic.json", "r03_pmc_hbm_traffic.json")
                        if os.path.exists(os.path.join(ROOT, "profiles", n))), os.path.join("profiles", "r06_pmc_hbm_traffic.json"))
            try:
                pmc = json.load(open(os.path.join(ROOT, src)))
                kb = 0.0
                for name, v in pmc["kernels"].items():
                    if any(p in name for p in pattern_names.get(dominant, ())):
                        kb += v["hbm_bytes_per_frame"]
                return (int(kb * frames_per_launch) if kb else None), src + " (rocprofv3 --pmc passes, corrected per MI355X_MICROARCH.md; not measured in this run)"
            except Exception:
                return None, None
        return fn

    if config == 2:
        from jxl_oxide_amd.synth import VardctWorkload
        stages = abi.STAGE_ALL

        def verify(ctx, frames, mine, wls, distinct):
            from oracle import pyoracle
            worst, ok, hist = 0, True, {}
            for d, wl in wls.items():
                f = frames[[i % distinct for i in mine].index(d)]
                got = ctx.download_result(f, stages)
                exp, _ = pyoracle.vardct_render(wl.desc(), stages, wl.width, wl.height)
                same = np.array_equal(got.view(np.uint32), exp.view(np.uint32))
                ok &= bool(same)
                if not same:
                    # distance in units in the last place: floats mapped to integers that order like the reals
                    def ordered(x):
                        i = x.view(np.int32).astype(np.int64)
                        return np.where(i < 0, -(i & 0x7fffffff), i)
                    dist = np.abs(ordered(got) - ordered(exp))
                    worst = max(worst, int(dist.max()))
                    edges = [0, 1, 2, 3, 5, 9, 17, 65, 1 << 62]
                    names = ["0", "1", "2", "3-4", "5-8", "9-16", "17-64", ">64"]
                    for nme, lo, hi in zip(names, edges[:-1], edges[1:]):
                        hist[nme] = hist.get(nme, 0) + int(((dist >= lo) & (dist < hi)).sum())
            res = {"ok": ok, "frames_checked": len(wls), "max_raw_bit_distance": worst,
                   "against": "oracle/ (C restatement of the reference's generic path), whole 3840x2160 frames, after the timed region"}
            if hist:
                res["ulp_histogram"] = hist
            if os.environ.get("JXLGPU_POST_FAST", "0") not in ("", "0"):
                res["post_fast"] = ("JXLGPU_POST_FAST=1: the NON-bit-exact post kernel (products with one refined reciprocal instead of the "
                                    "correctly rounded divisions, fma-contracted sums / polynomials); a measured option, never the default")
            return res

        def compulsory_bytes(wl, f):
            nnz = sum(int(np.count_nonzero(wl.coeff[c])) for c in range(3))
            nvb = int((wl.kind <= 26).sum())
            side = f.algorithmic_bytes(stages) - f.algorithmic_bytes(stages & ~abi.STAGE_TRANSFORM)   # = the coefficient planes 8(d) counts
            return f.algorithmic_bytes(stages) - side + nnz * 4 + nvb * 6

        def alg_bytes(f, group):
            npx, ncell = W4K * H4K, (W4K // 8) * (H4K // 8)
            if group == 1:
                return f.algorithmic_bytes(abi.STAGE_LF | abi.STAGE_TRANSFORM)
            if group == 2:
                return npx * 24 + ncell * 4
            return f.algorithmic_bytes(stages)

        return {
            "metric": "Megapixels/sec decoded (4K VarDCT d1)", "dtype": "f32", "batched": True,
            "input": {"grouped": "decoded state resident in HBM exactly as the entropy decoder emits it: per pass group the non_zeros counts and "
                                 "(dx, dy, coeff) triples of write_hf_coeff (4 B per non-zero coefficient), LF quant, block map; the transform "
                                 "kernels consume the lists directly — NO device-side layout pass (retile / zero-fill + scatter) exists on this path",
                      "dense_i32": "decoded state resident in HBM with the coefficients already in 8x8 cells: the per-frame retile pass of dense "
                                   "row-major planes (jxlgpu_vardct_upload) is NOT in the timed region (round-2 form of the bench)",
                      "sparse_i16": "as dense_i32 (zero-fill + scatter at upload, not timed)"}[transport],
            "workload": f"{W4K}x{H4K} VarDCT d1 XYB, Gabor + EPF iters 2, XYB->sRGB f32 planar (BASELINE config 2 frames, config 4 batch of 64)",
            "out_w": W4K, "out_h": H4K,
            "make": lambda d: VardctWorkload(W4K, H4K, seed=2000 + d, nz_fraction=nz),
            "upload": lambda ctx, wl: ctx.vardct_upload(wl.desc(coeff_transport=transport)),
            "render": lambda ctx, frames: ctx.vardct_render_batch(frames, stages),
            "groups": (1, 2),
            "group_names": {1: "transform: transform_items_batch_kernel<0..3> + transform_special_batch_kernel (V4-V8)",
                            2: "post: post_pk_batch_kernel (+ post_ring_batch_kernel beside it): Gabor + EPF steps 1,2 + XYB->sRGB"},
            "alg_bytes": alg_bytes, "verify": verify,
            "compulsory_bytes": compulsory_bytes if transport == "grouped" else None,
            "traffic": pmc_traffic({1: ("transform_",), 2: ("post_pk", "post_stream", "post_ring")}),
        }
    if config == 5:
        from jxl_oxide_amd.synth import VardctWorkload
        stages = abi.STAGE_ALL

        def verify(ctx, frames, mine, wls, distinct):
            from oracle import pyoracle
            d, wl = next(iter(wls.items()))
            f = frames[[i % distinct for i in mine].index(d)]
            got = ctx.download_result(f, stages)
            exp, _ = pyoracle.vardct_render(wl.desc(), stages, 2 * W4K, 2 * H4K)
            return {"ok": bool(np.array_equal(got.view(np.uint32), exp.view(np.uint32))), "frames_checked": 1,
                    "against": "oracle/, whole 7680x4320 output"}

        def render(ctx, frames):
            # V1-V8 of the frames share launches; the post stage (not the default pipeline) follows frame by frame
            ctx.vardct_render_batch(frames, stages)

        return {
            "metric": "Megapixels/sec decoded (8K out: coded 4K VarDCT, 2x upsampling, EPF iters 3, PQ)", "dtype": "f32", "batched": False,
            "workload": "coded 3840x2160 VarDCT, Gabor + EPF iters 3, 2x non-separable upsampling -> 7680x4320, intensity target 4000, Rec.2100 PQ (BASELINE config 5)",
            "out_w": 2 * W4K, "out_h": 2 * H4K,
            "make": lambda d: VardctWorkload(W4K, H4K, seed=5000 + d, epf_iters=3, upsampling=2, intensity_target=4000.0, hdr_pq=True, nz_fraction=nz),
            "upload": lambda ctx, wl: ctx.vardct_upload(wl.desc(coeff_transport=transport)),
            "render": render, "groups": (1, 2),
            "group_names": {1: "transform: V4-V8", 2: "post: fused_post_kernel<true,4> (Gabor + EPF step 0) + post_pk_kernel (steps 1, 2) + upsample2_lds_kernel (2x + PQ, packed colour chain)"},
            "alg_bytes": lambda f, g: W4K * H4K * 12 + 4 * W4K * H4K * 12,  # 12 B per coded px in, 12 B per output px out
            "verify": verify, "traffic": lambda d, n: (None, None),
        }
    from jxl_oxide_amd.synth_modular import ModularWorkload
    stages = abi.STAGE_ALL | abi.STAGE_MODULAR_TO_FLOAT
    W8K, H8K = 7680, 4320

    def verify(ctx, frames, mine, wls, distinct):
        from oracle import pyoracle
        d, wl = next(iter(wls.items()))
        f = frames[[i % distinct for i in mine].index(d)]
        got = f.ctx.modular_render(f, stages)
        exp = pyoracle.modular_render(wl.desc(), stages, wl.width, wl.height)
        return {"ok": bool(np.array_equal(got.view(np.uint32), exp.view(np.uint32))), "frames_checked": 1,
                "against": "oracle/, whole 7680x4320 frame (integer Squeeze inverse + float tail)"}

    def render(ctx, frames):
        for f in frames:   # (frames alternate between the job's contexts: `contexts` below)
            f.ctx.modular_render(f, stages, to_host=False)

    return {
        "metric": "Megapixels/sec decoded (8K Modular Squeeze lossy)", "dtype": "int16", "batched": False,
        "workload": "7680x4320 Modular, lossy Squeeze (22 steps) + self-correcting-predictor residuals (single-leaf MA tree) on the 67 carved "
                    "sub-channels, 16-bit buffers, XYB dequant + EPF iters 2 (sigma_for_modular) + XYB->sRGB (BASELINE config 3); whole frames, "
                    "one render call each, the frames of a step alternating between the job's contexts (JXLGPU_BENCH_CONTEXTS, default 3)",
        "out_w": W8K, "out_h": H8K,
        "make": lambda d: ModularWorkload(W8K, H8K, kind="squeeze", lossy=True, i16=True, epf_iters=2, seed=3 + d, residual=6),
        "upload": lambda ctx, wl: ctx.modular_upload(wl.desc()),
        # Frames alternate between JXLGPU_BENCH_CONTEXTS contexts (own streams each: one renderer per thread is the reference's caller
        # pattern, jxl-oxide-cli/src/decode.rs:293-304), so that the predictor pass of one frame — VALU issue, no memory — overlaps the
        # HBM-bound Squeeze / filter launches of others.  Round 6, 8-12 frames per step: 16.5-16.7 GP/s with one context, 16.3 with two
        # (the two frames stay in lockstep), 18.9-19.6 with three, 18.6-18.7 four, 19.8 five, 19.3-19.7 six / eight
        # (profiles/r06_predictor_experiments.txt).  With the round-3 predictor step the same spread gave 15.2 / 14.0 / 15.7.
        "contexts": int(os.environ.get("JXLGPU_BENCH_CONTEXTS", "3")),
        "render": render, "groups": (3, 2),
        "group_names": {3: "modular: predict_kernel (per carved sub-channel) + inverse Squeeze (segment-parallel kernels)", 2: "post: to_float + EPF + XYB->sRGB"},
        "alg_bytes": lambda f, g: W8K * H8K * (6 + 12),  # 3 x i16 in + 3 x f32 out per pixel
        "verify": verify, "traffic": lambda d, n: (None, None),
    }


def other_configs():
    """BASELINE configs 3 and 5 on this box, in the default line (N = 1): `bench.py --config C` as a child process on a short
    job (4 frames, one distinct workload, 5 timed steps), its value / verification / roofline fraction copied here."""
    res = {}
    for c in (3, 5):
        key = "config%d" % c
        try:
            r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--config", str(c), "--frames", "4", "--distinct", "1",
                                "--steps", "5", "--warmup", "2", "--no-cpu-baseline", "--no-extras"],
                               capture_output=True, text=True, timeout=300, env=dict(os.environ, JXLGPU_NO_CANARY="1"))
            d = json.loads(r.stdout.strip().splitlines()[-1])
            rf = d.get("roofline") or {}
            res[key] = {"metric": d.get("metric"), "value": d.get("value"), "unit": d.get("unit"), "ms_per_step": d.get("ms_per_step"),
                        "frames_per_step": (d.get("config") or {}).get("frames_per_gpu_per_step"), "dtype": d.get("dtype"),
                        "workload": (d.get("config") or {}).get("workload"),
                        "verified": d.get("verified"),
                        "roofline": {k: rf.get(k) for k in ("bound", "achieved", "peak", "unit", "frac", "kernel", "avg_launch_ms",
                                                            "pipeline_algorithmic_frac")},
                        "command": "python bench.py --config %d --frames 4 --distinct 1 --steps 5 --warmup 2" % c}
        except Exception as e:  # noqa: BLE001
            res[key] = {"error": "%s: %s" % (type(e).__name__, e)}
    return res


def end_to_end(ctx, wl, mp_per_frame):
    """PCIe-inclusive, what a decoder calling the library once per frame feels (never `value`):
    serial    : upload -> render -> u8 interleaved download into pageable memory, one frame at a time, best of 5, with
                the split (host build / H2D / render / format + D2H);
    pipelined : the same three calls per frame with three frames in flight — jxlgpu_vardct_upload does not wait for the
                device, jxlgpu_frame_format_output writes a pinned buffer (jxlgpu_host_alloc) asynchronously,
                jxlgpu_frame_wait + jxlgpu_frame_free retire the oldest frame: upload k+1 overlaps render k overlaps
                download k-1."""
    import numpy as np
    from jxl_oxide_amd import abi
    d = wl.desc(coeff_transport="grouped")
    for _ in range(3):  # warm the three staging buffers and the pool
        f = ctx.vardct_upload(d); ctx.vardct_render(f, abi.STAGE_ALL, to_host=False); ctx.format_output(f, abi.FMT_U8, 1); f.free()
    best, split = 1e9, None
    for _ in range(5):
        t0 = time.perf_counter()
        f = ctx.vardct_upload(d)
        t1 = time.perf_counter()
        ctx.vardct_render(f, abi.STAGE_ALL, to_host=False)
        ctx.synchronize()
        t2 = time.perf_counter()
        ref = ctx.format_output(f, abi.FMT_U8, 1)
        t3 = time.perf_counter()
        sp = ctx.upload_split()
        f.free()
        if t3 - t0 < best:
            best = t3 - t0
            split = {"upload_call_ms": round((t1 - t0) * 1e3, 3), "host_build_ms": round(sp[0], 3), "h2d_ms": round(sp[4], 3),
                     "render_ms": round((t2 - t1) * 1e3 - sp[4], 3), "format_d2h_pageable_ms": round((t3 - t2) * 1e3, 3)}
    depth, n = 3, 48
    outs = [ctx.host_alloc((wl.height, wl.width, 3), np.uint8) for _ in range(depth)]
    pipe = 1e9
    for _ in range(2):
        inflight = []
        t0 = time.perf_counter()
        for k in range(n):
            f = ctx.vardct_upload(d)
            ctx.vardct_render(f, abi.STAGE_ALL, to_host=False)
            ctx.format_output_async(f, abi.FMT_U8, outs[k % depth])
            inflight.append(f)
            if len(inflight) == depth:
                g = inflight.pop(0); ctx.frame_wait(g); g.free()
        for g in inflight:
            ctx.frame_wait(g); g.free()
        pipe = min(pipe, (time.perf_counter() - t0) / n)
    same = bool(np.array_equal(outs[(n - 1) % depth], ref))
    for o in outs:
        ctx.host_free(o)
    # Two calling threads, a context each (the reference's own caller pattern: keyframes rendered from a rayon par_iter,
    # jxl-oxide-cli/src/decode.rs:293-304; one jxlgpu_ctx per rendering thread): the host build of one thread's frame k + 1
    # overlaps the other thread's calls — the 0.45 ms work-list build is what a single caller serialises on.
    pipe2, same2 = None, None
    try:
        import threading
        from jxl_oxide_amd import runtime
        ncall = 2
        ctxs = [runtime.Context(ctx.device if hasattr(ctx, "device") else 0) for _ in range(ncall)]
        descs = [d] * ncall   # read-only for the library: both callers upload from the same decoded state
        outs2 = [[c.host_alloc((wl.height, wl.width, 3), np.uint8) for _ in range(depth)] for c in ctxs]
        errs = []

        def caller(i, frames_each, bar):
            try:
                c, dd = ctxs[i], descs[i]
                bar.wait(timeout=60)
                inflight = []
                for k in range(frames_each):
                    f = c.vardct_upload(dd)
                    c.vardct_render(f, abi.STAGE_ALL, to_host=False)
                    c.format_output_async(f, abi.FMT_U8, outs2[i][k % depth])
                    inflight.append(f)
                    if len(inflight) == depth:
                        g = inflight.pop(0); c.frame_wait(g); g.free()
                for g in inflight:
                    c.frame_wait(g); g.free()
            except Exception as e:  # noqa: BLE001
                errs.append(repr(e))

        for rep in range(3):   # the first repetition warms the second contexts' pools and staging buffers
            bar = threading.Barrier(ncall + 1)
            ths = [threading.Thread(target=caller, args=(i, n // ncall, bar)) for i in range(ncall)]
            for t in ths:
                t.start()
            bar.wait(timeout=60)
            t0 = time.perf_counter()
            for t in ths:
                t.join(timeout=300)
            dt = (time.perf_counter() - t0) / (n // ncall * ncall)
            if rep and not errs:
                pipe2 = dt if pipe2 is None else min(pipe2, dt)
        same2 = bool(not errs and all(np.array_equal(outs2[i][(n // ncall - 1) % depth], ref) for i in range(ncall)))
        for i, c in enumerate(ctxs):
            for o in outs2[i]:
                c.host_free(o)
            c.close()
        if errs:
            print(f"end_to_end: two-caller leg: {errs[:2]}", file=sys.stderr)
            pipe2 = None
    except Exception as e:  # noqa: BLE001 - a reporting extra never takes the line down
        print(f"end_to_end: two-caller leg failed: {e!r}", file=sys.stderr)
        pipe2 = None
    nzw = int(sum(int(d.hf_groups[g].num_nz) for g in range(d.num_hf_groups)))
    best_pipe, callers = (pipe2, 2) if (pipe2 is not None and same2 and pipe2 < pipe) else (pipe, 1)
    return {"ms_per_frame": round(best_pipe * 1e3, 3), "MP_per_s": round(mp_per_frame / best_pipe, 1), "callers": callers,
            "ms_per_frame_one_caller": round(pipe * 1e3, 3),
            "ms_per_frame_two_callers": None if pipe2 is None else round(pipe2 * 1e3, 3),
            "two_caller_output_identical_to_serial": same2,
            "serial_ms_per_frame": round(best * 1e3, 3), "serial_MP_per_s": round(mp_per_frame / best, 1), "serial_split": split,
            "pipelined_output_identical_to_serial": same,
            "what": "per frame: jxlgpu_vardct_upload (grouped non-zero lists; work-list build on the ctx's host threads, one pinned arena, "
                    "one async H2D) + jxlgpu_vardct_render + jxlgpu_frame_format_output (u8 interleaved).  ms_per_frame: three frames in "
                    "flight, output into pinned memory (upload k+1 / render k / download k-1 overlap) — from ONE calling thread "
                    "(ms_per_frame_one_caller) and from TWO calling threads with a context each, the reference CLI's pattern "
                    "(ms_per_frame_two_callers); ms_per_frame is the better of the two, `callers` says which; serial_*: one frame "
                    "at a time, output into pageable memory, best of 5",
            "list_MB_per_frame": None if nzw is None else round(nzw * 4 / 1e6, 2)}


def cpu_baseline(config, seconds):
    """oracle/cpu_bench.py in child processes: a few (frames in parallel) x (threads per frame) splits
    of this box's cores, `seconds` each; the best one is the baseline, the table is kept."""
    ncpu = len(os.sched_getaffinity(0))
    splits = [(1, min(16, ncpu))]
    for procs in (4, 16, 32):
        thr = max(1, min(16, ncpu // procs))
        if procs * thr <= ncpu and (procs, thr) not in splits:
            splits.append((procs, thr))
    # many single-frame workers with one or two threads each: the scalar port scales better across frames than inside
    # one (OpenMP fork / join per stage); bounded by the free memory of the box (a worker holds ~1.5 GB of a 4K frame)
    try:
        import psutil
        free_gb = psutil.virtual_memory().available / 2**30
    except Exception:
        free_gb = 64.0
    for procs, thr in ((64, 1), (64, 2), (128, 1)):
        if procs * thr <= ncpu and procs * 2.0 <= free_gb * 0.5 and (procs, thr) not in splits:
            splits.append((procs, thr))
    table, best = [], None
    for procs, thr in splits:
        try:
            r = subprocess.run([sys.executable, "-m", "oracle.cpu_bench", "--config", str(config), "--procs", str(procs),
                                "--threads", str(thr), "--seconds", str(seconds)], cwd=ROOT, capture_output=True, text=True,
                               timeout=300)
            j = json.loads(r.stdout.strip().splitlines()[-1])
            table.append(j)
            if best is None or j["MP_per_s"] > best["MP_per_s"]:
                best = j
        except Exception as e:  # a failed split must not cost the bench line
            table.append({"procs": procs, "threads": thr, "error": str(e)[:100]})
    if best is None:
        return None
    return {
        "value": best["MP_per_s"], "unit": "MP/s", "cores": best["cores"], "kind": "port",
        "MP_per_s_per_core": round(best["MP_per_s"] / max(best["cores"], 1), 3),
        "sample": f"{best['procs']} frames in parallel x {best['threads']} OpenMP threads each, whole frames of the same workload for "
                  f"{best['seconds']} s (oracle/: scalar C restatement of jxl-oxide's generic path, -O3 -march=native; the reference's "
                  "rayon workers run SSE/AVX2 code on x86, which would be faster still: this is NOT jxl-oxide's rayon speed)",
        "host_threads": ncpu, "splits": table,
    }


if __name__ == "__main__":
    main()
