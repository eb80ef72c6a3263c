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


class _FakeCtx:
    """Stands in for runtime.Context in the CPU test of shard.PeerWriteGather: "device memory" is a file every rank
    maps, the IPC handle its name.  What is under test is the PROTOCOL — who allocates, who opens, the probe, and above
    all that a failure on ONE rank makes EVERY rank raise (a rank left inside a collective would hang the job)."""
    BASE = 1 << 40

    def __init__(self, tmpdir, rank, fail=None):
        self.tmpdir, self.rank, self.fail = tmpdir, rank, fail
        self.mm = None

    def device_alloc(self, nbytes):
        import numpy as np
        if self.fail == "alloc":
            raise RuntimeError("simulated allocation failure")
        self.path = os.path.join(self.tmpdir, "buf.bin")
        self.mm = np.memmap(self.path, dtype=np.uint8, mode="w+", shape=(nbytes,))
        return self.BASE

    def device_free(self, ptr):
        self.mm = None

    def ipc_export(self, ptr):
        return self.path.encode().ljust(64, b"\0")[:64] if len(self.path) <= 64 else b"x" * 64

    def ipc_open(self, handle):
        import numpy as np
        if self.fail == "open":
            raise RuntimeError("simulated hipIpcOpenMemHandle failure")
        self.mm = np.memmap(os.path.join(self.tmpdir, "buf.bin"), dtype=np.uint8, mode="r+")
        return self.BASE

    def ipc_close(self, ptr):
        self.mm = None

    def device_upload(self, ptr, arr):
        import numpy as np
        if self.fail == "probe":
            return                      # the bytes silently go nowhere: the owner must notice
        a = np.ascontiguousarray(arr).view(np.uint8).ravel()
        self.mm[ptr - self.BASE:ptr - self.BASE + a.size] = a
        self.mm.flush()

    def device_download(self, ptr, shape, dtype):
        import numpy as np
        n = int(np.prod(shape)) * np.dtype(dtype).itemsize
        fresh = np.memmap(os.path.join(self.tmpdir, "buf.bin"), dtype=np.uint8, mode="r")
        return np.array(fresh[ptr - self.BASE:ptr - self.BASE + n]).view(dtype).reshape(shape)

    def format_output_to(self, frame, sample_format, dev_ptr, orientation=1):
        self.device_upload(dev_ptr, frame)          # a "frame" is its formatted bytes here
        return 0, 0

    def synchronize(self):
        pass


def _peer_worker(rank, world, port, tmpdir, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import numpy as np
    import torch.distributed as dist
    from jxl_oxide_amd import shard
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ok = True
    try:
        slot = 4096
        # (1) the happy path: rank r's two "frames" land in slots [r][0..1] of the owner's buffer
        ctx = _FakeCtx(tmpdir, rank)
        pw = shard.PeerWriteGather(ctx, slot, 2, dst=0)
        frames = [np.full(slot, 16 * rank + k + 1, dtype=np.uint8) for k in range(2)]
        pw.write(frames, 0)
        pw.finish()
        if rank == 0:
            got = pw.result()
            for r in range(world):
                for k in range(2):
                    ok &= bool((got[r, k] == 16 * r + k + 1).all())
        else:
            ok &= pw.result() is None
        pw.close()
        # (2) a failure on ONE rank — the owner's allocation, a writer's open, a probe that never arrives — must raise
        #     on EVERY rank (and leave nobody inside a collective: the barrier below would hang otherwise)
        for who, what in ((0, "alloc"), (1, "open"), (1, "probe")):
            ctx = _FakeCtx(tmpdir, rank, fail=what if rank == who else None)
            try:
                shard.PeerWriteGather(ctx, slot, 2, dst=0)
                ok = False
            except RuntimeError as e:
                ok &= "PeerWriteGather unavailable" in str(e)
            dist.barrier()
        q.put((rank, ok))
    finally:
        dist.destroy_process_group()


def test_peer_write_gather_protocol_and_collective_failures(tmp_path):
    """shard.PeerWriteGather's set-up under gloo with two processes and a fake context (no GPU): data lands where it
    should, and every single-rank failure turns into the same exception on all ranks."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_peer_worker, args=(r, 2, port, str(tmp_path), q)) for r in range(2)]
    for p in procs:
        p.start()
    results = dict(q.get(timeout=240) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert results == {0: True, 1: True}
