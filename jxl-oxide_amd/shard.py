"""Multi-GPU sharding of the render path (one process per GPU, torch.distributed; backend "nccl"
is RCCL over xGMI on MI355X, "gloo" in the CPU tests).

The reference has no distributed code at all; what shards is the work unit it hands to rayon:
  * frames are independent (BASELINE config 4): contiguous blocks of frames per rank, no
    communication on the data path;
  * inside one VarDCT frame, groups are independent through the inverse DCT
    (jxl-render/src/vardct/mod.rs:319; varblocks never cross a 256-px group,
    jxl-vardct/src/hf_metadata.rs:144-158) and the filters reach <= 7 px (+2 coded px for
    upsampling) across a boundary (jxl-render/src/util.rs:60-110).  A rank therefore renders a band
    of whole group rows extended by ONE halo group row on each side (recomputed, not exchanged:
    cheaper than a halo exchange) and keeps only its own rows — bit-identical to the
    unsharded render.
The only collective is the final gather of finished planes to the root (`gather_planes`).
"""
import numpy as np


def frame_shard(n_frames, rank, world):
    """Contiguous block of frame indices owned by `rank` (sizes differ by at most one)."""
    base, rem = divmod(n_frames, world)
    start = rank * base + min(rank, rem)
    return range(start, start + base + (1 if rank < rem else 0))


def band_plan(height, world, group_dim=256):
    """[(y0, y1, ext_y0, ext_y1)] per rank: rows [y0, y1) owned, [ext_y0, ext_y1) rendered.
    Bands are whole group rows; ranks beyond the number of group rows get empty bands."""
    n_rows = -(-height // group_dim)
    plan = []
    for r in range(world):
        rows = frame_shard(n_rows, r, world)
        if len(rows) == 0:
            plan.append((0, 0, 0, 0))
            continue
        y0, y1 = rows.start * group_dim, min(rows.stop * group_dim, height)
        ext0 = max(0, y0 - group_dim)
        ext1 = min(height, y1 + group_dim)
        plan.append((y0, y1, ext0, ext1))
    return plan


def band_rows(height, world, align=8):
    """[(y0, y1)] per rank: OUTPUT rows of a frame split into `world` bands of whole `align`-row units
    (BASELINE config 5: the groups of one frame sharded across the GPUs).  Each rank renders its band
    with jxlgpu_vardct_render_region: the library transforms the varblocks within reach of the band
    (<= 98 coded rows beyond it, whatever the band's size) — not the whole 256-row halo group rows that
    `band_plan` + `slice_vardct_band` (round 2, kept for the CPU tests) recompute."""
    units = -(-height // align)
    out = []
    for r in range(world):
        u = frame_shard(units, r, world)
        out.append((min(u.start * align, height), min(u.stop * align, height)))
    return out


def slice_vardct_band(wl, ext_y0, ext_y1):
    """A shallow copy of a synth.VardctWorkload restricted to coded rows [ext_y0, ext_y1)
    (multiples of the group size, except the frame's last row): every per-frame array is cut along
    y; geometry-free parameters are shared."""
    import copy
    assert ext_y0 % 256 == 0
    assert not wl.noise.enabled, "noise seeds depend on absolute group positions: shard noisy frames by frame, not by band"
    b = copy.copy(wl)
    b.height = ext_y1 - ext_y0
    c0, c1 = ext_y0 // 8, -(-ext_y1 // 8)
    b.h8 = c1 - c0
    b.hr = b.h8 * 8
    b.coeff = np.ascontiguousarray(wl.coeff[:, c0 * 8:c1 * 8, :])
    b.kind = np.ascontiguousarray(wl.kind[c0:c1])
    b.hf_mul = np.ascontiguousarray(wl.hf_mul[c0:c1])
    b.sigma = np.ascontiguousarray(wl.sigma[c0:c1])
    b.lfq = [np.ascontiguousarray(p[c0:c1]) for p in wl.lfq]
    t0, t1 = ext_y0 // 64, -(-ext_y1 // 64)
    b.xfy = np.ascontiguousarray(wl.xfy[t0:t1])
    b.bfy = np.ascontiguousarray(wl.bfy[t0:t1])
    b._keep = []
    # LF groups are 2048 px: a band that starts inside one keeps the per-LF-group extra_precision
    # of the rows it came from (desc() derives it from the LF group index)
    b.lf_group_row0 = ext_y0 // (wl.group_dim * 8)
    return b


def gather_planes(local, dst=0, group=None):
    """Gathers equally-shaped tensors (one per rank) to `dst`.  Returns the list on `dst`, None
    elsewhere.  One collective; RCCL when the tensors live on GPUs."""
    import torch.distributed as dist
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    if world == 1:
        return [local]
    bufs = [local.new_empty(local.shape) for _ in range(world)] if rank == dst else None
    dist.gather(local, bufs, dst=dst, group=group)
    return bufs


def render_frames_sharded(n_frames, render_fn, rank, world):
    """Each rank renders its block of frames with `render_fn(frame_index) -> array/tensor`."""
    return {i: render_fn(i) for i in frame_shard(n_frames, rank, world)}


def gather_formatted(ctx, frame, sample_format, orientation=1, dst=0, group=None):
    """The last render of `frame`, formatted on the device (interleaved, oriented, u8 / u16 / f32:
    jxlgpu_frame_format_output) straight into a torch tensor on this rank's
    GPU, then gathered to `dst` with ONE collective (RCCL over xGMI when the tensors are on GPUs).
    u8 output moves 3 B/px instead of the 12 B/px of planar f32.  Import torch and touch the GPU
    before creating `ctx` (README: one HIP runtime per process).  Returns the list of per-rank
    (h, w, 3) tensors on `dst`, None elsewhere."""
    import ctypes as C

    import torch

    from . import abi
    w, h = frame.out_size(abi.STAGE_ALL)
    ow, oh = (w, h) if orientation <= 4 else (h, w)
    dt = {abi.FMT_F32: torch.float32, abi.FMT_U16: torch.uint16, abi.FMT_U8: torch.uint8}[sample_format]
    local = torch.empty((oh, ow, 3), dtype=dt, device="cuda")
    fmt = abi.FormatDesc(sample_format, orientation)
    rw, rh = C.c_uint32(), C.c_uint32()
    ctx._check(ctx.lib.jxlgpu_frame_format_output(ctx.handle, frame.handle, C.byref(fmt), local.data_ptr(),
                                                  abi.MEM_DEVICE, C.byref(rw), C.byref(rh)))
    ctx.synchronize()
    return gather_planes(local, dst=dst, group=group)


def torch_dtype_of(sample_format):
    import torch

    from . import abi
    return {abi.FMT_F32: torch.float32, abi.FMT_U16: torch.uint16, abi.FMT_U8: torch.uint8}[sample_format]


def format_frames_into(ctx, frames, sample_format, local, orientation=1, first_slot=0):
    """The last render of every frame (whole frames or regions of the same size), formatted on the device
    (jxlgpu_frame_format_output) into slots first_slot, first_slot + 1, ... of the (slots, h, w, 3) tensor
    `local` on this rank's GPU.  Asynchronous on the library's stream."""
    import ctypes as C

    from . import abi
    fmt = abi.FormatDesc(sample_format, orientation)
    rw, rh = C.c_uint32(), C.c_uint32()
    step = local[0].numel() * local.element_size()
    assert first_slot + len(frames) <= local.shape[0]
    for i, f in enumerate(frames):
        ctx._check(ctx.lib.jxlgpu_frame_format_output(ctx.handle, f.handle, C.byref(fmt), local.data_ptr() + (first_slot + i) * step,
                                                      abi.MEM_DEVICE, C.byref(rw), C.byref(rh)))
        # a result shorter than the slot (the last band of a frame) fills the slot's first rows
        assert rw.value == local.shape[2] and rh.value <= local.shape[1], "results must have the slot's width and fit its height"
        # (the tensor may be a byte view of wider samples: (slots, h, w, 3 * sample_size) uint8)


def gather_formatted_batch(ctx, frames, sample_format, orientation=1, dst=0, group=None, slots=None):
    """Config 4's stitched output: every frame of this rank formatted on the device into ONE tensor
    (slots, h, w, 3), then ONE gather to `dst` (RCCL over xGMI for world > 1; nothing to move for
    world == 1).  Frames must have the same size.  `slots`: frames per rank rounded up over the job
    (ranks own blocks that differ by one frame when the batch does not divide: every rank gathers the
    same shape, the unused slot stays empty); default: all_reduce(max) of the local counts.  A rank
    without frames passes `slots` and a frame size through `like`... it cannot know one: the bench
    gives every rank at least one frame or uses PipelinedGather with an explicit shape.
    Returns the list of per-rank tensors on `dst` (None elsewhere)."""
    import torch
    import torch.distributed as dist

    from . import abi
    multi = dist.is_initialized() and dist.get_world_size(group) > 1
    if slots is None:
        slots = len(frames)
        if multi:
            t = torch.tensor([slots], dtype=torch.int64, device="cuda")
            dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
            slots = int(t.item())
    if not frames:
        raise ValueError("gather_formatted_batch: this rank owns no frame (give the gather an explicit shape: PipelinedGather)")
    w, h = frames[0].out_size(abi.STAGE_ALL)
    ow, oh = (w, h) if orientation <= 4 else (h, w)
    local = torch.zeros((max(slots, len(frames)), oh, ow, 3), dtype=torch_dtype_of(sample_format), device="cuda")
    format_frames_into(ctx, frames, sample_format, local, orientation)
    ctx.synchronize()
    if not multi:
        return [local]
    return gather_planes(local, dst=dst, group=group)


class PipelinedGather:
    """One tensor per step from every rank to `dst`, overlapped with the next step's rendering.

    The library renders and formats on its own HIP stream; RCCL works on its own stream behind torch's
    current stream.  Per step:  slot(step) -> the local tensor to format into (its previous gather is
    waited for ON THE LIBRARY'S STREAM, not by the host);  submit(step) -> an event on the library's
    stream makes torch's stream wait for the formatting, the gather is issued with async_op=True, and
    an event behind it guards the buffer.  The host never blocks, so step k + 1's kernels run while
    step k's bytes cross xGMI.  Two buffer sets; finish() drains.  On CPU tensors (gloo, the tests) the
    same calls run synchronously.  VERDICT r2 item 5a."""

    def __init__(self, shape, dtype, device, lib_stream=None, dst=0, group=None, depth=2, force_collective=False):
        import torch
        import torch.distributed as dist
        self.dist, self.torch = dist, torch
        self.group, self.dst = group, dst
        # force_collective: issue the gather even in a one-rank group (exercises the stream / event plumbing on one GPU)
        self.multi = dist.is_initialized() and (dist.get_world_size(group) > 1 or force_collective)
        self.rank = dist.get_rank(group) if self.multi else 0
        self.world = dist.get_world_size(group) if self.multi else 1
        # `dst` is a GLOBAL rank (what dist.gather takes); with a sub-group the group rank differs from it
        self.global_rank = dist.get_rank() if self.multi else 0
        self.is_dst = (self.global_rank == dst) if self.multi else True
        self.depth = depth
        self.cuda = torch.device(device).type == "cuda"
        self.local = [torch.zeros(shape, dtype=dtype, device=device) for _ in range(depth)]
        self.recv = [None] * depth
        if self.multi and self.is_dst:
            self.recv = [[torch.zeros(shape, dtype=dtype, device=device) for _ in range(self.world)] for _ in range(depth)]
        self.ext = torch.cuda.ExternalStream(lib_stream) if (self.cuda and lib_stream) else None
        self.done = [None] * depth   # event behind the gather that last read local[i]
        self.work = [None] * depth
        self.bytes_to_dst = 0

    def slot(self, step):
        i = step % self.depth
        if self.done[i] is not None and self.ext is not None:
            self.ext.wait_event(self.done[i])      # the formatting kernels queue behind the gather that read this buffer
        elif self.done[i] is not None:
            # no library stream to order behind: the host waits for the gather that last read this buffer, or the next
            # formatting (on a stream torch knows nothing about) would overwrite it mid-flight
            self.done[i].synchronize()
        elif self.work[i] is not None and not self.cuda:
            self.work[i].wait()
        return self.local[i]

    def submit(self, step):
        i = step % self.depth
        if not self.multi:
            return
        torch, dist = self.torch, self.dist
        if self.ext is not None:
            ev = torch.cuda.Event()
            ev.record(self.ext)
            torch.cuda.current_stream().wait_event(ev)
        w = dist.gather(self.local[i], self.recv[i] if self.is_dst else None, dst=self.dst, group=self.group,
                        async_op=True)
        self.work[i] = w
        if self.cuda:
            w.wait()                                # torch's stream (not the host) waits for the collective
            d = torch.cuda.Event()
            d.record(torch.cuda.current_stream())
            self.done[i] = d
        else:
            w.wait()
        if self.is_dst:
            self.bytes_to_dst += (self.world - 1) * self.local[i].numel() * self.local[i].element_size()

    def finish(self, step=None):
        """Drains; returns the per-rank tensors of `step` on dst (own slot = local copy), else None."""
        if self.cuda:
            self.torch.cuda.synchronize()
        if step is None or not self.multi:
            return [self.local[(step or 0) % self.depth]] if step is not None else None
        i = step % self.depth
        if not self.is_dst:
            return None
        return self.recv[i]


class PeerWriteGather:
    """The stitched output WITHOUT a collective: rank `dst` owns world x slots x slot_bytes of
    its HBM (jxlgpu_device_alloc) and exports it; every other rank maps it (jxlgpu_ipc_open: a peer mapping over
    xGMI) and its formatting kernels store straight into its own slots there — all ranks at once, each over its own
    link to `dst`, where a rooted `dist.gather` funnels one ring through rank 0's receive path.  The host never
    waits inside a step; a step is complete on `dst` after every rank's stream has drained (finish()).  Only the 64
    handle bytes and the barriers go through torch.distributed (any backend: gloo is enough)."""

    def __init__(self, ctx, slot_bytes, slots, dst=0, group=None):
        import torch.distributed as dist
        self.dist, self.ctx, self.group, self.dst = dist, ctx, group, dst
        self.multi = dist.is_initialized() and dist.get_world_size(group) > 1
        self.rank = dist.get_rank(group) if self.multi else 0
        self.world = dist.get_world_size(group) if self.multi else 1
        self.global_rank = dist.get_rank() if self.multi else 0
        self.is_dst = (self.global_rank == dst) if self.multi else True
        self.slot_bytes, self.slots = int(slot_bytes), int(slots)
        self.bytes_to_dst = 0
        self.base = None
        self.owned = False
        # Every step below is collective-safe: a rank that fails still takes part in the exchanges, and ALL ranks
        # raise together (the caller falls back to another gather on every rank, or on none).
        import numpy as np
        err = None
        obj = [None]
        if self.is_dst:
            try:
                self.base = ctx.device_alloc(self.world * self.slots * self.slot_bytes)
                self.owned = True
                obj = [ctx.ipc_export(self.base)] if self.multi else [None]
            except Exception as e:  # noqa: BLE001
                err = f"owner: {type(e).__name__}: {e}"
                obj = [None]
        if self.multi:
            dist.broadcast_object_list(obj, src=dst, group=group)
            if not self.is_dst:
                if obj[0] is None:
                    err = "the owner could not allocate / export the buffer"
                else:
                    try:
                        self.base = ctx.ipc_open(obj[0])
                    except Exception as e:  # noqa: BLE001
                        err = f"rank {self.rank}: {type(e).__name__}: {e}"
            # probe: every writer puts a pattern at the start of its first slot THROUGH the mapping, the owner reads it back
            probe = np.full(64, 0xA0 + (self.rank & 0xF), dtype=np.uint8)
            if err is None and not self.is_dst and self.slot_bytes >= 64:
                try:
                    ctx.device_upload(self.base + self.rank * self.slots * self.slot_bytes, probe)
                except Exception as e:  # noqa: BLE001
                    err = f"rank {self.rank} probe write: {type(e).__name__}: {e}"
            errs = [None] * self.world
            dist.all_gather_object(errs, err, group=group)
            if self.is_dst and not any(errs) and self.slot_bytes >= 64:
                try:   # dst must reach the verdict broadcast whatever happens here: the other ranks are waiting in it
                    got = ctx.device_download(self.base, (self.world, self.slots * self.slot_bytes), np.uint8)
                    for r in range(self.world):
                        if r != self.rank and not (got[r, :64] == 0xA0 + (r & 0xF)).all():
                            err = f"probe pattern of rank {r} did not arrive through its peer mapping"
                    for r in range(self.world):   # the probe bytes are not output: back to zero
                        ctx.device_upload(self.base + r * self.slots * self.slot_bytes, np.zeros(64, dtype=np.uint8))
                except Exception as e:  # noqa: BLE001
                    err = f"owner's probe read-back: {type(e).__name__}: {e}"
            verdict = [err if self.is_dst else None]
            dist.broadcast_object_list(verdict, src=dst, group=group)
            errs = [e for e in errs if e] + ([verdict[0]] if verdict[0] else [])
            if errs:
                self._abort()
                raise RuntimeError("PeerWriteGather unavailable: " + "; ".join(sorted(set(errs))))
        elif err:
            raise RuntimeError("PeerWriteGather unavailable: " + err)
        self.mine = self.base + self.rank * self.slots * self.slot_bytes

    def _abort(self):
        """Local clean-up after a failed set-up (no collective: every rank is on its way out)."""
        try:
            if self.base is not None and self.multi and not self.is_dst:
                self.ctx.ipc_close(self.base)
            if self.owned and self.base is not None:
                self.ctx.device_free(self.base)
        except Exception:  # noqa: BLE001
            pass
        self.base = None

    def write(self, frames, sample_format, orientation=1, first_slot=0):
        """Formats the last render of every frame into this rank's slots first_slot, ... in dst's buffer (asynchronous)."""
        assert first_slot + len(frames) <= self.slots
        for i, f in enumerate(frames):
            self.ctx.format_output_to(f, sample_format, self.mine + (first_slot + i) * self.slot_bytes, orientation)
        if not self.is_dst:
            self.bytes_to_dst += len(frames) * self.slot_bytes

    def finish(self):
        """Every rank's stores have landed on dst when this returns on all ranks."""
        self.ctx.synchronize()
        if self.multi:
            self.dist.barrier(group=self.group)

    def result(self):
        """On dst: the stitched bytes as a (world, slots, slot_bytes) uint8 array (host copy); None elsewhere."""
        import numpy as np
        if not self.is_dst:
            return None
        return self.ctx.device_download(self.base, (self.world, self.slots, self.slot_bytes), np.uint8)

    def close(self):
        if self.base is None:
            return
        if self.multi and not self.is_dst:
            self.ctx.ipc_close(self.base)
        if self.multi:
            self.dist.barrier(group=self.group)   # nobody maps the buffer any more
        if self.owned:
            self.ctx.device_free(self.base)
        self.base = None
