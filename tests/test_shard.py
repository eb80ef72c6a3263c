"""Sharding logic (CPU): frame blocks, group-row bands with a recomputed halo, and the gather —
world_size 2 over gloo.  The per-rank renderer here is the oracle (tests may use it as the
checker); on the GPU box the same code paths run with the HIP library (tests/test_gpu_shard.py)."""
import os
import socket

import numpy as np
import pytest

from jxl_oxide_amd import abi, shard
from jxl_oxide_amd.synth import VardctWorkload


def test_frame_shard_partitions():
    for n in (0, 1, 7, 8, 64, 65):
        for world in (1, 2, 3, 8):
            got = [i for r in range(world) for i in shard.frame_shard(n, r, world)]
            assert got == list(range(n))
            sizes = [len(shard.frame_shard(n, r, world)) for r in range(world)]
            assert max(sizes) - min(sizes) <= 1


def test_band_plan_covers_frame():
    for h in (100, 256, 700, 2160, 4320):
        for world in (1, 2, 4, 8):
            plan = shard.band_plan(h, world)
            rows = [y for (y0, y1, _, _) in plan for y in range(y0, y1)]
            assert rows == list(range(h))
            for (y0, y1, e0, e1) in plan:
                if y1 > y0:
                    assert e0 <= y0 and e1 >= y1 and e0 % 256 == 0
                    assert y0 - e0 in (0, 256) and (e1 - y1 in (0, 256) or e1 == h)


def test_band_render_is_bit_identical_to_full_frame(oracle):
    """Group-row bands + one recomputed halo group row == unsharded render (EPF iters 3: 7-px reach)."""
    wl = VardctWorkload(300, 1100, seed=31, epf_iters=3)
    full, _ = oracle.vardct_render(wl.desc(), abi.STAGE_ALL, wl.width, wl.height)
    for world in (2, 3):
        out = np.zeros_like(full)
        for (y0, y1, e0, e1) in shard.band_plan(wl.height, world):
            if y1 == y0:
                continue
            band = shard.slice_vardct_band(wl, e0, e1)
            r, _ = oracle.vardct_render(band.desc(), abi.STAGE_ALL, band.width, band.height)
            out[:, y0:y1] = r[:, y0 - e0:y1 - e0]
        assert np.array_equal(out.view(np.uint32), full.view(np.uint32))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    import torch
    import torch.distributed as dist
    from oracle import pyoracle
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        # (1) frame sharding: 5 frames over 2 ranks, gather to root
        n_frames = 5
        wls = {i: VardctWorkload(72, 40, seed=500 + i, epf_iters=1) for i in range(n_frames)}

        def render(i):
            return pyoracle.vardct_render(wls[i].desc(), abi.STAGE_ALL, 72, 40)[0]
        mine = shard.render_frames_sharded(n_frames, render, rank, world)
        slots = -(-n_frames // world)
        local = torch.zeros((slots, 3, 40, 72), dtype=torch.float32)
        for k, i in enumerate(sorted(mine)):
            local[k] = torch.from_numpy(mine[i])
        gathered = shard.gather_planes(local, dst=0)
        ok = True
        if rank == 0:
            for r in range(world):
                for k, i in enumerate(shard.frame_shard(n_frames, r, world)):
                    ok &= bool(np.array_equal(gathered[r][k].numpy(), render(i)))
        # (2) band sharding of one frame + gather + stitch
        wl = VardctWorkload(200, 700, seed=77, epf_iters=2)
        plan = shard.band_plan(wl.height, world)
        y0, y1, e0, e1 = plan[rank]
        band = shard.slice_vardct_band(wl, e0, e1)
        r = pyoracle.vardct_render(band.desc(), abi.STAGE_ALL, band.width, band.height)[0]
        max_rows = max(p[1] - p[0] for p in plan)
        local = torch.zeros((3, max_rows, wl.width), dtype=torch.float32)
        local[:, :y1 - y0] = torch.from_numpy(r[:, y0 - e0:y1 - e0])
        gathered = shard.gather_planes(local, dst=0)
        if rank == 0:
            full = pyoracle.vardct_render(wl.desc(), abi.STAGE_ALL, wl.width, wl.height)[0]
            stitched = np.zeros_like(full)
            for rr, (a, b, _, _) in enumerate(plan):
                stitched[:, a:b] = gathered[rr][:, :b - a].numpy()
            ok &= bool(np.array_equal(stitched.view(np.uint32), full.view(np.uint32)))
        # (3) the gather moves formatted (interleaved u8) frames just as well: 3 B/px instead of 12
        #     (on the GPU the formatting is jxlgpu_frame_format_output, shard.gather_formatted)
        mine_u8 = {i: pyoracle.format_output(v, abi.FMT_U8, 1) for i, v in mine.items()}
        local = torch.zeros((slots, 40, 72, 3), dtype=torch.uint8)
        for k, i in enumerate(sorted(mine_u8)):
            local[k] = torch.from_numpy(mine_u8[i])
        gathered = shard.gather_planes(local, dst=0)
        if rank == 0:
            for r in range(world):
                for k, i in enumerate(shard.frame_shard(n_frames, r, world)):
                    ok &= bool(np.array_equal(gathered[r][k].numpy(), pyoracle.format_output(render(i), abi.FMT_U8, 1)))
        # (4) shard.PipelinedGather (the overlapped gather of bench.py at N > 1), here on CPU tensors: three steps,
        #     two buffer sets, uneven frame blocks (an empty slot on the rank that owns one frame less)
        pg = shard.PipelinedGather((slots, 40, 72, 3), torch.uint8, "cpu", dst=0)
        last = None
        for step in range(3):
            buf = pg.slot(step)
            buf.zero_()
            for k, i in enumerate(sorted(mine_u8)):
                buf[k] = torch.from_numpy(mine_u8[i]) + step      # a different payload every step
            pg.submit(step)
            last = step
        got = pg.finish(last)
        if rank == 0:
            for r in range(world):
                for k, i in enumerate(shard.frame_shard(n_frames, r, world)):
                    exp = (pyoracle.format_output(render(i), abi.FMT_U8, 1).astype(np.int32) + last).astype(np.uint8)
                    ok &= bool(np.array_equal(got[r][k].numpy(), exp))
        ok &= shard.band_rows(4320, 8)[0] == (0, 544) and shard.band_rows(4320, 8)[-1][1] == 4320
        ok &= [b for b in shard.band_rows(100, 3)] == [(0, 40), (40, 72), (72, 100)]
        q.put((rank, ok))
    finally:
        dist.destroy_process_group()


def test_two_rank_gloo_shard_and_gather(oracle):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = dict(q.get(timeout=240) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert results == {0: True, 1: True}
