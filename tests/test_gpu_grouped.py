"""JXLGPU_COEFF_GROUPED: the transform kernels fed by the decoder's per-varblock non-zero lists
(`non_zeros` + (dx, dy, coeff) triples of jxl-vardct/src/hf_coeff.rs:188-254) instead of dense
coefficient planes.  The oracle always sees the dense i32 planes of the same workload; results must
be bit-identical (same arithmetic, only the staging differs)."""
import ctypes as C

import numpy as np
import pytest

from jxl_oxide_amd import abi
from jxl_oxide_amd.synth import VardctWorkload

pytestmark = pytest.mark.gpu

S_TR = abi.STAGE_LF | abi.STAGE_TRANSFORM
S_ALL = abi.STAGE_ALL


def _render(ctx, wl, stages, transport="grouped"):
    f = ctx.vardct_upload(wl.desc(coeff_transport=transport))
    try:
        return ctx.vardct_render(f, stages)
    finally:
        f.free()


def _same(got, exp, what):
    assert got.shape == exp.shape, what
    if not np.array_equal(got.view(np.uint32), exp.view(np.uint32)):
        a, b = got.view(np.int32).astype(np.int64), exp.view(np.int32).astype(np.int64)
        bad = np.argwhere(a != b)
        raise AssertionError(f"{what}: {bad.shape[0]} samples differ, max raw-bit distance {np.abs(a - b).max()}, first {bad[0]}")


@pytest.mark.parametrize("t", list(range(27)))
def test_each_transform_type_from_lists(gpu_ctx, oracle, t):
    """Every TransformType (the >= 128-px ones go through the dense fallback built from the lists)."""
    bw, bh = abi.DCT_SELECT_SIZE[t]
    w = min(max(64, bw * 8 * 2 + 8), 256 + 64)
    h = min(max(64, bh * 8 * 2 + 8), 256 + 64)
    wl = VardctWorkload(w, h, seed=100 + t, types=[t], zero_fraction=0.5)
    assert (wl.kind == t).any()
    exp, _ = oracle.vardct_render(wl.desc(), S_TR, w, h)
    _same(_render(gpu_ctx, wl, S_TR), exp, abi.TRANSFORM_NAMES[t])


@pytest.mark.parametrize("size", [(8, 8), (9, 7), (255, 257), (520, 300), (1040, 600)])
@pytest.mark.parametrize("zero_fraction", [0.85, 0.3, 0.0])
def test_mixed_frames_all_stages(gpu_ctx, oracle, size, zero_fraction):
    """The cfg-2 shape mix at several sizes and densities: 0.0 keeps every coefficient (lists longer
    than one scatter round of every shape class)."""
    w, h = size
    wl = VardctWorkload(w, h, seed=7, zero_fraction=zero_fraction)
    exp, _ = oracle.vardct_render(wl.desc(), S_ALL, w, h)
    _same(_render(gpu_ctx, wl, S_ALL), exp, f"{size} zeros={zero_fraction}")


def test_large_coefficients_take_the_division_path(gpu_ctx, oracle):
    """|q| >= 256 leaves the quant_bias_numerator / k table: the row is redone with the division."""
    wl = VardctWorkload(264, 200, seed=11)
    rng = np.random.default_rng(5)
    sel = rng.random(wl.coeff.shape) < 0.002
    wl.coeff[sel] = rng.integers(-30000, 30000, size=int(sel.sum()))
    # LLF positions are not coded in HF (hf_coeff.rs): keep them empty as the generator does
    ys, xs = np.nonzero(wl.kind <= 26)
    for cy, cx in zip(ys, xs):
        bw, bh = abi.DCT_SELECT_SIZE[int(wl.kind[cy, cx])]
        wl.coeff[:, cy * 8:cy * 8 + bh, cx * 8:cx * 8 + bw] = 0
    exp, _ = oracle.vardct_render(wl.desc(), S_TR, wl.width, wl.height)
    _same(_render(gpu_ctx, wl, S_TR), exp, "large coefficients")


def test_dense_fallback_and_batch_match(gpu_ctx, oracle, monkeypatch):
    """JXLGPU_NO_SPARSE_TR expands the lists to dense cells (the dense kernels); batched launches of
    list-fed frames; a batch mixing list-fed and dense frames (rendered one by one)."""
    from jxl_oxide_amd import runtime
    wls = [VardctWorkload(520, 264, seed=1), VardctWorkload(300, 520, seed=2), VardctWorkload(264, 200, seed=3)]
    exps = [oracle.vardct_render(w.desc(), S_ALL, w.width, w.height)[0] for w in wls]
    monkeypatch.setenv("JXLGPU_NO_SPARSE_TR", "1")
    ctx2 = runtime.Context(0)
    try:
        for w, e in zip(wls, exps):
            _same(_render(ctx2, w, S_ALL), e, "dense fallback")
    finally:
        ctx2.close()
    monkeypatch.delenv("JXLGPU_NO_SPARSE_TR")
    for transports in (["grouped"] * 3, ["grouped", "dense_i32", "grouped"]):
        frames = [gpu_ctx.vardct_upload(w.desc(coeff_transport=t)) for w, t in zip(wls, transports)]
        try:
            for _ in range(2):
                gpu_ctx.vardct_render_batch(frames, S_ALL)
                gpu_ctx.synchronize()
                for w, f, e in zip(wls, frames, exps):
                    _same(gpu_ctx.download_result(f), e, f"batch {transports}")
        finally:
            for f in frames:
                f.free()


def test_malformed_lists_are_rejected_or_ignored(gpu_ctx, oracle):
    wl = VardctWorkload(264, 200, seed=4)
    # counts that do not add up to num_nz
    d = wl.desc(coeff_transport="grouped")
    d.hf_groups[0].num_nz += 1
    with pytest.raises(Exception) as e:
        gpu_ctx.vardct_upload(d)
    assert e.value.code == abi.ERR_INVALID_ARG
    # fewer varblocks than the block map holds
    d = wl.desc(coeff_transport="grouped")
    d.hf_groups[0].num_varblocks -= 1
    with pytest.raises(Exception) as e:
        gpu_ctx.vardct_upload(d)
    assert e.value.code == abi.ERR_INVALID_ARG
    # wrong number of groups
    d = wl.desc(coeff_transport="grouped")
    d.num_hf_groups += 1
    with pytest.raises(Exception) as e:
        gpu_ctx.vardct_upload(d)
    assert e.value.code == abi.ERR_INVALID_ARG
    # a position outside its varblock is ignored (memory-safe), everything else is unaffected
    wl8 = VardctWorkload(64, 64, seed=5, types=[0])
    d = wl8.desc(coeff_transport="grouped")
    keep = wl8._keep  # the list arrays `d` points into (the next desc() call replaces wl8._keep)
    hg = d.hf_groups[0]
    words = np.ctypeslib.as_array(hg.nz, shape=(hg.num_nz,))
    k = int(np.flatnonzero((words & 0xFF) < 8)[0])
    dx, dy = int(words[k] & 0xFF), int((words[k] >> 8) & 0xFF)
    words[k] = (words[k] & np.uint32(0xFFFF0000)) | np.uint32(200 | (dy << 8))  # dx = 200 in an 8x8 block
    # the same frame with that coefficient absent
    counts = np.ctypeslib.as_array(hg.nz_count, shape=(hg.num_varblocks * 3,)).astype(np.int64)
    owner = int(np.searchsorted(np.cumsum(counts), k, side="right"))
    vb, slot = divmod(owner, 3)
    ys, xs = np.nonzero(wl8.kind <= 26)
    order = np.lexsort((xs, ys))
    cy, cx = int(ys[order][vb]), int(xs[order][vb])
    wl8.coeff[(1, 0, 2)[slot], cy * 8 + dy, cx * 8 + dx] = 0
    exp, _ = oracle.vardct_render(wl8.desc(), S_TR, 64, 64)
    f = gpu_ctx.vardct_upload(d)
    try:
        _same(gpu_ctx.vardct_render(f, S_TR), exp, "out-of-block entry ignored")
    finally:
        f.free()
    del keep


def test_negative_quant_bias_takes_the_dense_kernels(gpu_ctx, oracle):
    """A zero coefficient times a negative quant_bias is -0.0, which the list-fed kernels (zeros never
    touched) would not produce: such a frame is expanded to dense cells at upload.  Same result as the oracle."""
    wl = VardctWorkload(264, 200, seed=21)
    d0 = wl.desc()
    d0.quant_bias[0] = -0.9
    exp, _ = oracle.vardct_render(d0, S_TR, wl.width, wl.height)
    d = wl.desc(coeff_transport="grouped")
    d.quant_bias[0] = -0.9
    f = gpu_ctx.vardct_upload(d)
    try:
        _same(gpu_ctx.vardct_render(f, S_TR), exp, "negative quant_bias")
    finally:
        f.free()


@pytest.mark.parametrize("transport", ["grouped", "dense_i32", "sparse_i16"])
def test_truncated_stream_allow_partial(gpu_ctx, oracle, transport):
    """`allow_partial` (jxl-render/src/vardct/mod.rs:275-305): the stream of some pass groups ends early; what was
    decoded stays, the varblocks never reached have no HF coefficients and are still dequantised + transformed with
    their LF.  Grouped lists then hold fewer varblocks than the block map; without the flag that is an error."""
    from jxl_oxide_amd.runtime import JxlGpuError
    wl = VardctWorkload(700, 520, seed=21, zero_fraction=0.5)   # 3 x 3 pass groups
    partial = {1: 0, 4: 37, 8: 5, 5: 10 ** 6}                   # nothing / part / part / everything decoded
    exp, _ = oracle.vardct_render(wl.desc(partial=partial), S_ALL, wl.width, wl.height)
    full, _ = oracle.vardct_render(wl.desc(), S_ALL, wl.width, wl.height)
    assert not np.array_equal(exp, full)
    f = gpu_ctx.vardct_upload(wl.desc(coeff_transport=transport, partial=partial))
    try:
        _same(gpu_ctx.vardct_render(f, S_ALL), exp, f"partial, {transport}")
    finally:
        f.free()
    if transport == "grouped":
        d = wl.desc(coeff_transport="grouped", partial=partial)
        d.allow_partial = 0
        with pytest.raises(JxlGpuError) as e:
            gpu_ctx.vardct_upload(d)
        assert e.value.code == abi.ERR_INVALID_ARG


@pytest.mark.parametrize("transport", ["grouped", "dense_i32"])
@pytest.mark.parametrize("size", [(264, 200), (2100, 300)])
def test_lf_frame_replaces_the_lf_stages(gpu_ctx, oracle, transport, size):
    """frame_header.flags.use_lf_frame (vardct/mod.rs:175-179): the LF image is handed over as f32 XYB planes; V1-V3
    do not run (the lf_quant pointers may be NULL), everything downstream reads those planes."""
    w, h = size
    wl = VardctWorkload(w, h, seed=31, lf_frame=True)
    d0 = wl.desc()
    exp, exp_lf = oracle.vardct_render(d0, S_ALL, w, h, want_lf=True, w8=wl.w8, h8=wl.h8)
    assert np.array_equal(exp_lf, wl.lf_frame[:, :, :wl.w8])
    d = wl.desc(coeff_transport=transport)
    for g in range(d.num_lf_groups):
        for c in range(3):
            d.lf_groups[g].lf_quant[c] = None
    f = gpu_ctx.vardct_upload(d)
    try:
        _same(gpu_ctx.vardct_render(f, S_ALL), exp, f"lf_frame {transport}")
        _same(gpu_ctx.download_lf(f, wl.w8, wl.h8), exp_lf, "LF image == the LF frame")
        # batched entry point and region render on the same frame
        gpu_ctx.vardct_render_batch([f], S_ALL)
        gpu_ctx.synchronize()
        _same(gpu_ctx.download_result(f), exp, "lf_frame, batch entry point")
        reg = (40, 24, min(160, w - 40), min(120, h - 24))
        got = gpu_ctx.vardct_render_region(f, S_ALL, reg)
        _same(got, np.ascontiguousarray(exp[:, reg[1]:reg[1] + reg[3], reg[0]:reg[0] + reg[2]]), "lf_frame, region")
    finally:
        f.free()


@pytest.mark.parametrize("shifts", [[2, 0], [4, 1, 0], [1, 0, 0]])
@pytest.mark.parametrize("size", [(264, 200), (520, 300)])
def test_progressive_passes_accumulate(gpu_ctx, oracle, size, shifts):
    """Multi-pass (progressive) frames through the grouped transport: one list set per (pass, group), pass p adding
    `unpack_signed(ucoeff) << coeff_shift` to what the earlier passes left (hf_coeff.rs:234-235).  The oracle sees
    the summed dense planes: the result must be bit-identical; so must a batched render of the frame."""
    w, h = size
    wl = VardctWorkload(w, h, seed=23, nz_fraction=0.2)
    exp, _ = oracle.vardct_render(wl.desc(), S_ALL, w, h)
    f = gpu_ctx.vardct_upload(wl.desc(coeff_transport="grouped", pass_shifts=shifts))
    try:
        _same(gpu_ctx.vardct_render(f, S_ALL), exp, f"{size} passes {shifts}")
        gpu_ctx.vardct_render_batch([f], S_ALL)
        gpu_ctx.synchronize()
        _same(gpu_ctx.download_result(f), exp, f"{size} passes {shifts}, batched")
    finally:
        f.free()


@pytest.mark.parametrize("shifts", [[2, 0], [3, 1, 0]])
def test_truncated_progressive_stream(gpu_ctx, oracle, shifts):
    """`allow_partial` together with several passes (the case partial decoding exists for: a progressive stream cut
    short): every (pass, group) section may end after any number of varblocks, later passes typically earlier.  The
    oracle sees the sum of the truncated parts."""
    import copy
    w, h = 600, 520            # 3 x 3 pass groups
    wl = VardctWorkload(w, h, seed=29, nz_fraction=0.2)
    last = len(shifts) - 1
    partial = {(0, 4): 61, (last, 0): 0, (last, 1): 17, (last, 4): 5, (last, 8): 10 ** 6}
    if last > 1:
        partial.update({(1, 1): 40, (1, 7): 0})
    wl_sum = copy.copy(wl)
    wl_sum.coeff = wl.progressive_truncated_coeff(shifts, partial)
    exp, _ = oracle.vardct_render(wl_sum.desc(), S_ALL, w, h)
    full, _ = oracle.vardct_render(wl.desc(), S_ALL, w, h)
    assert not np.array_equal(exp.view(np.uint32), full.view(np.uint32)), "the truncation removed nothing"
    f = gpu_ctx.vardct_upload(wl.desc(coeff_transport="grouped", pass_shifts=shifts, partial=partial))
    try:
        _same(gpu_ctx.vardct_render(f, S_ALL), exp, f"truncated progressive {shifts}")
        gpu_ctx.vardct_render_batch([f], S_ALL)
        gpu_ctx.synchronize()
        _same(gpu_ctx.download_result(f), exp, f"truncated progressive {shifts}, batched")
    finally:
        f.free()
    # without allow_partial the same lists are refused
    from jxl_oxide_amd.runtime import JxlGpuError
    d = wl.desc(coeff_transport="grouped", pass_shifts=shifts, partial=partial)
    d.allow_partial = 0
    with pytest.raises(JxlGpuError):
        gpu_ctx.vardct_upload(d)


def test_pass_lists_are_validated(gpu_ctx):
    from jxl_oxide_amd.runtime import JxlGpuError
    wl = VardctWorkload(264, 200, seed=5)
    d = wl.desc(coeff_transport="grouped", pass_shifts=[2, 0])
    d.hf_groups[d.num_hf_groups].num_nz += 1   # the second pass of group 0 no longer matches its counts
    with pytest.raises(JxlGpuError):
        gpu_ctx.vardct_upload(d)
    d = wl.desc(coeff_transport="grouped", pass_shifts=[2, 0])
    d.num_passes = 12
    with pytest.raises(JxlGpuError):
        gpu_ctx.vardct_upload(d)
