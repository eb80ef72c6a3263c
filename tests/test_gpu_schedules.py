"""The tuning switches of the batched render (csrc/common.h `Tuning`) change WHERE and WHEN launches run, never what they
compute: under every one of them a batch must give the oracle's bits.  Also: the Modular predictor waves on the side stream
(JXLGPU_PRED_LATE_STEPS), the guard allocator's 4-byte mode, and the switch that used to be allowed to differ — JXLGPU_POST_FAST,
since round 6 a compile-time option that the shipped library does not have."""
import numpy as np
import pytest

from jxl_oxide_amd import abi, runtime
from jxl_oxide_amd.synth import VardctWorkload
from jxl_oxide_amd.synth_modular import ModularWorkload

pytestmark = pytest.mark.gpu

# sizes with several strips / segments of the streaming post kernel and every transform family in the draw
_WLS = [(776, 520, 31), (520, 776, 32), (264, 200, 33)]


def _render_batch(ctx, wls, copies):
    frames = [ctx.vardct_upload(w.desc(coeff_transport="grouped")) for w in wls for _ in range(copies)]
    try:
        for _ in range(2):   # twice: the second pass runs behind the first one's events
            ctx.vardct_render_batch(frames, abi.STAGE_ALL)
        ctx.synchronize()
        return [ctx.download_result(f) for f in frames]
    finally:
        for f in frames:
            f.free()


@pytest.fixture(scope="module")
def expected(oracle):
    wls = [VardctWorkload(w, h, seed=s, nz_fraction=0.15) for w, h, s in _WLS]
    exp = [oracle.vardct_render(w.desc(), abi.STAGE_ALL, w.width, w.height)[0] for w in wls]
    return wls, exp


@pytest.mark.parametrize("env", [
    {},                                                   # the default schedule
    {"JXLGPU_BATCH_STREAM_ROWS": "96"},                   # round-4 segment height
    {"JXLGPU_BATCH_STREAM_ROWS": "32"},
    {"JXLGPU_BATCH_CHUNK": "4"},                          # several chunks per batch: transform(k+1) beside post(k)
    {"JXLGPU_BATCH_CHUNK": "4", "JXLGPU_BATCH_HEAVY": "24"},
    {"JXLGPU_BATCH_CHUNK": "4", "JXLGPU_BATCH_HEAVY": "8", "JXLGPU_RING_MODE": "1"},
    {"JXLGPU_BATCH_CHUNK": "4", "JXLGPU_BATCH_HEAVY": "31"},
    {"JXLGPU_BATCH_CHUNK": "4", "JXLGPU_RING_MODE": "2"},
    {"JXLGPU_BATCH_CHUNK": "4", "JXLGPU_TR_STREAMS": "5"},
    {"JXLGPU_BATCH_CHUNK": "4", "JXLGPU_TR_STREAMS": "3", "JXLGPU_TR_SIDE_MAX": "0"},
    {"JXLGPU_BATCH_CHUNK": "4", "JXLGPU_TR_SIDE_MAX": "0"},
    {"JXLGPU_BATCH_CHUNK": "4", "JXLGPU_BATCH_TR_MULT": "2"},                       # transform launches of two chunks, post launches of one
    {"JXLGPU_BATCH_CHUNK": "2", "JXLGPU_BATCH_TR_MULT": "4", "JXLGPU_TR_SIDE_MAX": "32"},
    {"JXLGPU_NO_BATCH_OVERLAP": "1"},
    {"JXLGPU_PK_TB": "1"},                                # round 6: the top / bottom image rows inside the streaming kernel instead of through ring tiles
    {"JXLGPU_PK_TB": "1", "JXLGPU_BATCH_CHUNK": "4"},
    {"JXLGPU_PK_TB": "1", "JXLGPU_BATCH_STREAM_ROWS": "32"},
    {"JXLGPU_BATCH_LF_MODE": "1", "JXLGPU_BATCH_CHUNK": "4"},   # LF launches one chunk ahead / behind the previous chunk's small families (round-6 experiments)
    {"JXLGPU_BATCH_LF_MODE": "2", "JXLGPU_BATCH_CHUNK": "4"},
    {"JXLGPU_STREAM_PRIO": "-1", "JXLGPU_BATCH_CHUNK": "4"},
], ids=lambda e: ",".join(f"{k[7:]}={v}" for k, v in e.items()) or "default")
def test_every_schedule_gives_the_same_bits(expected, monkeypatch, env):
    wls, exp = expected
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    ctx = runtime.Context(0)
    try:
        got = _render_batch(ctx, wls, copies=3)   # 9 frames: with chunks of 4, three launches per stage
    finally:
        ctx.close()
    for i, g in enumerate(got):
        e = exp[i // 3]
        assert np.array_equal(g.view(np.uint32), e.view(np.uint32)), (env, _WLS[i // 3])


def test_post_fast_cannot_be_switched_on_in_the_shipped_library(expected, monkeypatch):
    """JXLGPU_POST_FAST selected a non-bit-exact post kernel in round 5 (a measured option: +6.6 %, 42 % of the samples beyond
    1 ULP).  Since round 6 that kernel is compiled only with -DJXL_ENABLE_POST_FAST (tools/build_variant.sh): the library the
    tests and the bench load ignores the variable — no environment can change the bits."""
    wls, exp = expected
    for v in ("1", "0"):
        monkeypatch.setenv("JXLGPU_POST_FAST", v)
        ctx = runtime.Context(0)
        try:
            got = _render_batch(ctx, wls[:1], copies=1)[0]
        finally:
            ctx.close()
        assert np.array_equal(got.view(np.uint32), exp[0].view(np.uint32)), v


@pytest.mark.parametrize("late", ["0", "1", "3", "9"])
def test_predictor_waves_on_the_side_stream(oracle, monkeypatch, late):
    """Residuals of the first JXLGPU_PRED_LATE_STEPS forward Squeeze steps are predicted on a side stream while the deep
    levels are already being un-squeezed; the first inverse step that reads them waits.  Any split gives the same samples."""
    monkeypatch.setenv("JXLGPU_PRED_LATE_STEPS", late)
    ctx = runtime.Context(0)
    try:
        for (w, h, i16) in ((1100, 700, True), (523, 517, False)):
            wl = ModularWorkload(w, h, kind="squeeze", lossy=True, xyb=True, residual=6, i16=i16, seed=11)
            d = wl.desc()
            exp = oracle.modular_inverse(d, wl.shapes(), wl.dtype)
            f = ctx.modular_upload(d)
            try:
                for _ in range(2):
                    got = ctx.modular_inverse(f, wl.shapes(), wl.dtype)
                for c in range(len(exp)):
                    assert np.array_equal(got[c], exp[c]), (late, w, h, c)
            finally:
                f.free()
    finally:
        ctx.close()


def test_guard_mode_3(expected, monkeypatch):
    """JXLGPU_GUARD=3: every device buffer ends at an unmapped page with its size rounded up to 4 bytes only."""
    wls, exp = expected
    monkeypatch.setenv("JXLGPU_GUARD", "3")
    ctx = runtime.Context(0)
    try:
        got = _render_batch(ctx, wls[2:], copies=2)
    finally:
        ctx.close()
    for g in got:
        assert np.array_equal(g.view(np.uint32), exp[2].view(np.uint32))
