"""Sharding on the device: band-sharded renders through the HIP library reassemble to the
unsharded HIP render bit-for-bit, and a frame batch split over ranks equals the serial batch."""
import numpy as np
import pytest

from jxl_oxide_amd import abi, shard
from jxl_oxide_amd.synth import VardctWorkload

pytestmark = pytest.mark.gpu


def _render(gpu_ctx, wl, stages=abi.STAGE_ALL):
    f = gpu_ctx.vardct_upload(wl.desc())
    try:
        return gpu_ctx.vardct_render(f, stages)
    finally:
        f.free()


@pytest.mark.parametrize("world", [2, 3, 8])
def test_band_sharding_matches_full_frame(gpu_ctx, oracle, world):
    wl = VardctWorkload(520, 1300, seed=61, epf_iters=2)
    full = _render(gpu_ctx, wl)
    out = np.zeros_like(full)
    for (y0, y1, e0, e1) in shard.band_plan(wl.height, world):
        if y1 == y0:
            continue
        r = _render(gpu_ctx, shard.slice_vardct_band(wl, e0, e1))
        out[:, y0:y1] = r[:, y0 - e0:y1 - e0]
    assert np.array_equal(out.view(np.uint32), full.view(np.uint32))
    exp, _ = oracle.vardct_render(wl.desc(), abi.STAGE_ALL, wl.width, wl.height)
    assert np.array_equal(full.view(np.uint32), exp.view(np.uint32))


def test_frame_sharding_covers_batch(gpu_ctx):
    wls = [VardctWorkload(136, 72, seed=90 + i) for i in range(5)]
    serial = [_render(gpu_ctx, w) for w in wls]
    for world in (2, 4):
        got = {}
        for rank in range(world):
            got.update(shard.render_frames_sharded(len(wls), lambda i: _render(gpu_ctx, wls[i]), rank, world))
        assert sorted(got) == list(range(len(wls)))
        for i in range(len(wls)):
            assert np.array_equal(got[i], serial[i])


@pytest.mark.parametrize("world", [2, 3, 8])
@pytest.mark.parametrize("case", [dict(epf_iters=2), dict(epf_iters=3, upsampling=2, intensity_target=4000.0, hdr_pq=True)])
def test_region_bands_reassemble_to_the_full_frame(gpu_ctx, world, case):
    """BASELINE config 5's sharding: one uploaded frame, every rank's band of OUTPUT rows rendered with
    jxlgpu_vardct_render_region (shard.band_rows), formatted to u16 on the device, stitched == the whole
    frame's formatted render.  No halo group rows: the library transforms what the band's filters reach."""
    wl = VardctWorkload(520, 1300, seed=62, **case)
    f = gpu_ctx.vardct_upload(wl.desc(coeff_transport="grouped"))
    try:
        gpu_ctx.vardct_render(f, abi.STAGE_ALL, to_host=False)
        full = gpu_ctx.format_output(f, abi.FMT_U16, 1)
        H, W = full.shape[0], full.shape[1]
        out = np.zeros_like(full)
        for (y0, y1) in shard.band_rows(H, world):
            if y1 == y0:
                continue
            gpu_ctx.vardct_render_region(f, abi.STAGE_ALL, (0, y0, W, y1 - y0), to_host=False)
            out[y0:y1] = gpu_ctx.format_output(f, abi.FMT_U16, 1)
        assert np.array_equal(out, full)
    finally:
        f.free()


def _nccl_worker(rank, world, port, q):
    """One process per GPU: frames sharded over the ranks, rendered through the HIP library, formatted on
    the device and gathered to rank 0 with shard.PipelinedGather over RCCL, three steps in flight."""
    import os
    import torch
    import torch.distributed as dist
    from jxl_oxide_amd import runtime
    from oracle import pyoracle
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", rank))
    ok = True
    try:
        ctx = runtime.Context(rank)
        n_frames = 2 * world + 1     # uneven blocks: one rank has an empty slot
        wls = [VardctWorkload(264, 200, seed=700 + i) for i in range(n_frames)]
        mine = list(shard.frame_shard(n_frames, rank, world))
        frames = [ctx.vardct_upload(wls[i].desc(coeff_transport="grouped")) for i in mine]
        slots = -(-n_frames // world)
        pg = shard.PipelinedGather((slots, 200, 264, 3), torch.uint8, "cuda", lib_stream=ctx.stream(), dst=0,
                                   force_collective=(world == 1))
        for step in range(3):
            ctx.vardct_render_batch(frames, abi.STAGE_ALL)
            buf = pg.slot(step)
            shard.format_frames_into(ctx, frames, abi.FMT_U8, buf)
            pg.submit(step)
        got = pg.finish(2)
        if rank == 0:
            for r in range(world):
                for k, i in enumerate(shard.frame_shard(n_frames, r, world)):
                    exp, _ = pyoracle.vardct_render(wls[i].desc(), abi.STAGE_ALL, 264, 200)
                    ok &= bool(np.array_equal(got[r][k].cpu().numpy(), pyoracle.format_output(exp, abi.FMT_U8, 1)))
        for f in frames:
            f.free()
        ctx.close()
        q.put((rank, ok))
    finally:
        dist.destroy_process_group()


def _run_nccl(world):
    import socket
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mpc = mp.get_context("spawn")
    q = mpc.Queue()
    procs = [mpc.Process(target=_nccl_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = dict(q.get(timeout=600) for _ in procs)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert results == {r: True for r in range(world)}


def test_pipelined_gather_plumbing_on_one_gpu(oracle):
    """The stream / event plumbing of shard.PipelinedGather (library stream wrapped as a torch ExternalStream,
    async RCCL gather, buffer-reuse guard) in a one-rank nccl group: runs on a single MI355X."""
    _run_nccl(1)


def test_two_rank_nccl_shard_and_gather(oracle):
    """The gloo test of tests/test_shard.py over RCCL on two GPUs (skipped on one-GPU boxes)."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    _run_nccl(2)


def _peer_worker(rank, world, port, q):
    """Two PROCESSES on ONE GPU: the IPC export / open / peer-store path of shard.PeerWriteGather is the same code
    that runs between two GPUs (the mapping is a same-device one here); control traffic over gloo."""
    import os
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import numpy as np
    import torch.distributed as dist
    from jxl_oxide_amd import abi, runtime, shard
    from jxl_oxide_amd.synth import VardctWorkload
    from oracle import pyoracle
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ctx = runtime.Context(0)
        n_frames = 5
        wls = [VardctWorkload(264, 200, seed=40 + i) for i in range(n_frames)]
        mine = list(shard.frame_shard(n_frames, rank, world))
        frames = [ctx.vardct_upload(wls[i].desc(coeff_transport="grouped")) for i in mine]
        slots = -(-n_frames // world)
        pw = shard.PeerWriteGather(ctx, 200 * 264 * 3, slots, dst=0)
        for _ in range(3):   # steps overwrite the same slots, as the bench's passes do
            ctx.vardct_render_batch(frames, abi.STAGE_ALL)
            pw.write(frames, abi.FMT_U8)
        pw.finish()
        ok = True
        if rank == 0:
            got = pw.result().reshape(world, slots, 200, 264, 3)
            for r in range(world):
                for k, i in enumerate(shard.frame_shard(n_frames, r, world)):
                    exp, _ = pyoracle.vardct_render(wls[i].desc(), abi.STAGE_ALL, 264, 200)
                    ok &= bool(np.array_equal(got[r, k], pyoracle.format_output(exp, abi.FMT_U8, 1)))
        pw.close()
        for f in frames:
            f.free()
        ctx.close()
        q.put((rank, ok))
    finally:
        dist.destroy_process_group()


def test_peer_write_gather_two_processes_one_gpu(oracle):
    """jxlgpu_device_alloc / jxlgpu_ipc_export / jxlgpu_ipc_open + format_output into the mapping: rank 1's frames
    arrive in rank 0's buffer without a collective, bit-identical to the oracle's formatted renders."""
    import socket
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mpc = mp.get_context("spawn")
    q = mpc.Queue()
    procs = [mpc.Process(target=_peer_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = dict(q.get(timeout=600) for _ in procs)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert results == {0: True, 1: True}
