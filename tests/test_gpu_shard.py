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
