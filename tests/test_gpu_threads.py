"""The C ABI under concurrent callers (SURVEY §8b "Threading": keyframes are rendered from arbitrary
rayon workers, jxl-oxide-cli/src/decode.rs:293-304; one jxlgpu_ctx per rendering thread, no global
state in the library).  ctypes releases the GIL for the duration of a foreign call, so Python threads
do drive libjxlgpu.so concurrently."""
import threading

import numpy as np
import pytest

from jxl_oxide_amd import abi, runtime
from jxl_oxide_amd.synth import VardctWorkload
from jxl_oxide_amd.synth_modular import ModularWorkload

pytestmark = pytest.mark.gpu


def test_four_threads_four_contexts(gpu_ctx, oracle):
    """4 threads x own context, each uploading and rendering different frames (VarDCT dense / grouped
    transports, batched launches, a Modular frame) 6 times over, all compared with the oracle."""
    jobs = [
        ("vardct", VardctWorkload(520, 264, seed=31), "dense_i32"),
        ("vardct", VardctWorkload(300, 520, seed=32), "grouped"),
        ("batch", VardctWorkload(264, 200, seed=33), "grouped"),
        ("modular", ModularWorkload(333, 200, kind="squeeze", lossy=True, i16=True, epf_iters=2, seed=34, residual=6), None),
    ]
    stages_m = abi.STAGE_ALL | abi.STAGE_MODULAR_TO_FLOAT
    expected = []
    for kind, wl, _ in jobs:
        if kind == "modular":
            expected.append(oracle.modular_render(wl.desc(), stages_m, wl.width, wl.height))
        else:
            expected.append(oracle.vardct_render(wl.desc(), abi.STAGE_ALL, wl.width, wl.height)[0])
    errors = []
    barrier = threading.Barrier(len(jobs))

    def work(i):
        try:
            kind, wl, transport = jobs[i]
            ctx = runtime.Context(0)
            try:
                barrier.wait(timeout=60)
                for rep in range(6):
                    if kind == "modular":
                        f = ctx.modular_upload(wl.desc())
                        got = ctx.modular_render(f, stages_m)
                        f.free()
                    elif kind == "batch":
                        frames = [ctx.vardct_upload(wl.desc(coeff_transport=transport)) for _ in range(3)]
                        ctx.vardct_render_batch(frames, abi.STAGE_ALL)
                        ctx.synchronize()
                        got = ctx.download_result(frames[rep % 3])
                        for f in frames:
                            f.free()
                    else:
                        f = ctx.vardct_upload(wl.desc(coeff_transport=transport))
                        got = ctx.vardct_render(f, abi.STAGE_ALL)
                        f.free()
                    if not np.array_equal(got.view(np.uint32), expected[i].view(np.uint32)):
                        errors.append(f"thread {i} ({kind}) repetition {rep}: result differs from the oracle")
            finally:
                ctx.close()
        except Exception as e:  # noqa: BLE001 - reported to the main thread
            errors.append(f"thread {i}: {type(e).__name__}: {e}")

    threads = [threading.Thread(target=work, args=(i,)) for i in range(len(jobs))]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=600)
    assert not any(t.is_alive() for t in threads), "a rendering thread hung"
    assert not errors, errors


def test_two_contexts_interleaved_on_one_thread(oracle):
    """Two contexts used alternately from one thread: uploads, renders and frees interleaved; neither
    sees the other's state (separate streams, pools, staging buffers)."""
    a, b = runtime.Context(0), runtime.Context(0)
    try:
        wa, wb = VardctWorkload(264, 200, seed=41), VardctWorkload(520, 300, seed=42, epf_iters=3)
        ea = oracle.vardct_render(wa.desc(), abi.STAGE_ALL, wa.width, wa.height)[0]
        eb = oracle.vardct_render(wb.desc(), abi.STAGE_ALL, wb.width, wb.height)[0]
        fa = a.vardct_upload(wa.desc(coeff_transport="grouped"))
        fb = b.vardct_upload(wb.desc())
        a.vardct_render(fa, abi.STAGE_ALL, to_host=False)
        b.vardct_render(fb, abi.STAGE_ALL, to_host=False)
        fa2 = a.vardct_upload(wa.desc())
        gb = b.download_result(fb)
        ga = a.download_result(fa)
        ga2 = a.vardct_render(fa2, abi.STAGE_ALL)
        fb.free()
        gb2 = b.vardct_render_host(wb.desc(), abi.STAGE_ALL, wb.width, wb.height)
        fa.free()
        fa2.free()
        for got, exp in ((ga, ea), (ga2, ea), (gb, eb), (gb2, eb)):
            assert np.array_equal(got.view(np.uint32), exp.view(np.uint32))
    finally:
        a.close()
        b.close()
