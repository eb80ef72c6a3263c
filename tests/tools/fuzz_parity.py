#!/usr/bin/env python3
"""Randomised geometry sweep, device against oracle, for the kernels whose work decomposition depends on the frame's
shape (lane-packed predictor pass, lane-per-piece Squeeze, region renders, list-fed transforms): sizes, sample type,
transform chain, predictor, filters drawn at random; every mismatch is printed with the parameters that reproduce it.

    python tests/tools/fuzz_parity.py SECONDS [SEED]          (on an MI355X; uses the oracle: test infrastructure)
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402

from jxl_oxide_amd import abi, runtime  # noqa: E402
from jxl_oxide_amd.synth import VardctWorkload, make_extra_channel  # noqa: E402
from jxl_oxide_amd.synth_modular import ModularWorkload  # noqa: E402
from oracle import pyoracle  # noqa: E402


def same_bits(a, b):
    """Bit-identical; where the expected value is NaN (HLG op lists: a negative luminance mix has no real power,
    jxl-color/src/tf.rs:118-143) a NaN of any payload."""
    a, b = np.ascontiguousarray(a), np.ascontiguousarray(b)
    if a.shape != b.shape:
        return False
    if a.dtype != np.float32:
        return bool(np.array_equal(a, b))
    an, bn = np.isnan(a), np.isnan(b)
    return bool(np.array_equal(an, bn) and np.array_equal(np.where(an, np.float32(0), a).view(np.uint32),
                                                          np.where(bn, np.float32(0), b).view(np.uint32)))


# colour op lists besides the default XYB -> sRGB (synth.configure_color): (mode, intensity targets it may be drawn with)
COLOUR_MODES = [("tone_map_srgb", (1000.0, 4000.0)), ("tone_map_min_nits", (600.0, 10000.0)), ("bt709", (255.0,)), ("clip_p3_dci", (255.0,)),
                ("gamma22", (255.0,)), ("hlg", (255.0, 300.0, 400.0, 1000.0, 4000.0)), ("pq_to_hlg", (400.0, 4000.0, 10000.0)),
                ("pq_to_hlg_1000", (999.0, 1000.0, 1001.0))]


def modular_case(rng):
    kind = rng.choice(["squeeze", "squeeze", "squeeze", "palette", "gray", "raw", "lossless_rgb8", "ycbcr420", "ycbcr422"])
    w = int(rng.choice([rng.integers(1, 40), rng.integers(40, 300), rng.integers(300, 1100)]))
    h = int(rng.choice([rng.integers(1, 40), rng.integers(40, 300), rng.integers(300, 700)]))
    kw = dict(kind=str(kind), i16=bool(rng.integers(0, 2)), seed=int(rng.integers(0, 1000)))
    if kind == "squeeze":
        kw["lossy"] = bool(rng.integers(0, 2))
        kw["xyb"] = kw["lossy"]
        if rng.random() < 0.6:
            kw["residual"] = int(rng.choice([0, 1, 4, 5, 6, 6, 6, 7, 9, 12, 13]))
    if kind in ("palette", "gray") and rng.random() < 0.5:
        kw["residual"] = int(rng.choice([5, 6, 13]))
    if kind.startswith("ycbcr"):
        kw["xyb"] = False
        w, h = max(w, 2), max(h, 2)
    if kind == "raw":
        w, h = max(w, 9), max(h, 9)
    if kind in ("squeeze", "palette", "gray") and rng.random() < 0.4:
        # group_size_shift 0 / 2 / 3 (jxl-frame/src/header.rs:299-301); with 1024 sometimes a size that has subgrids wider than 512
        kw["group_dim"] = int(rng.choice([128, 512, 1024, 1024]))
        if kw["group_dim"] == 1024 and rng.random() < 0.5:
            w, h = int(rng.integers(513, 2300)), int(rng.integers(300, 1300))
    # round 6: per-unit leaves (a leaf of its own per decode unit; "axis": trees that split on y / x inside the unit) and
    # Squeeze chains that squeeze residual channels again
    if kind in ("squeeze", "palette") and "residual" not in kw and rng.random() < 0.35:
        kw["leaves"] = str(rng.choice(["mixed", "axis", "axis"]))
        if kind == "squeeze":
            kw["lossy"], kw["xyb"] = False, False
    if kind == "squeeze" and not kw.get("lossy", True) and w >= 40 and h >= 40 and rng.random() < 0.5:
        kw["squeeze_plan"] = RESQUEEZE_PLANS[int(rng.integers(0, len(RESQUEEZE_PLANS)))]
    return w, h, kw


RESQUEEZE_PLANS = [
    [[(1, 1, 0, 3), (0, 1, 0, 3), (1, 1, 0, 3), (0, 1, 9, 3), (1, 1, 9, 3), (0, 1, 0, 3)]],
    [[(1, 1, 0, 3), (0, 1, 0, 3)], None],
    [[(1, 0, 1, 2), (0, 0, 1, 2), (1, 1, 0, 7), (0, 1, 0, 7), (1, 1, 3, 4)]],
    [None, [(0, 1, 3, 2), (1, 1, 3, 2)]],
]


def predictor_case(rng):
    """Plain RGB8 planes behind one predictor pass: whole-channel and 256 x 256 units, every leaf kind."""
    w = int(rng.choice([rng.integers(1, 40), rng.integers(40, 300), rng.integers(300, 1100)]))
    h = int(rng.choice([rng.integers(1, 40), rng.integers(40, 300), rng.integers(300, 700)]))
    kw = dict(kind="predictor", i16=bool(rng.integers(0, 2)), seed=int(rng.integers(0, 1000)))
    r = rng.random()
    if r < 0.6:
        kw["leaves"] = str(rng.choice(["mixed", "axis", "axis"]))
    else:
        kw["predictor"] = int(rng.integers(0, 14))
        # (the generator's weighted-predictor forward has no leaf offset)
        kw["pred_offset"] = 0 if kw["predictor"] == 6 else int(rng.choice([0, 0, -3, 7]))
    if rng.random() < 0.3:
        kw["group_dim"] = int(rng.choice([128, 512, 1024]))
    return w, h, kw


def run_modular(ctx, rng):
    w, h, kw = predictor_case(rng) if rng.random() < 0.15 else modular_case(rng)
    wl = ModularWorkload(w, h, **kw)
    d = wl.desc()
    if kw["kind"].startswith("ycbcr"):
        stages = abi.STAGE_ALL | abi.STAGE_MODULAR_TO_FLOAT
        exp = pyoracle.modular_render(d, stages, w, h)
        f = ctx.modular_upload(d)
        try:
            got = ctx.modular_render(f, stages)
        finally:
            f.free()
        ok = np.array_equal(got.view(np.uint32), exp.view(np.uint32))
    else:
        exp = pyoracle.modular_inverse(d, wl.shapes(), wl.dtype)
        f = ctx.modular_upload(d)
        try:
            got = ctx.modular_inverse(f, wl.shapes(), wl.dtype)
        finally:
            f.free()
        ok = all(np.array_equal(g, e) for g, e in zip(got, exp))
    return ok, ("modular", w, h, kw)


def run_vardct(ctx, rng):
    w = int(rng.integers(8, 900))
    h = int(rng.integers(8, 600))
    kw = dict(seed=int(rng.integers(0, 1000)), epf_iters=int(rng.integers(0, 4)), gabor=bool(rng.integers(0, 2)),
              zero_fraction=float(rng.choice([0.0, 0.5, 0.85, 0.97])))
    if rng.random() < 0.25:
        kw["upsampling"] = int(rng.choice([2, 4, 8]))
        w, h = min(w, 300), min(h, 200)
    if rng.random() < 0.3:
        mode, its = COLOUR_MODES[int(rng.integers(0, len(COLOUR_MODES)))]
        kw["color_mode"], kw["intensity_target"] = mode, float(rng.choice(its))
    wl = VardctWorkload(w, h, **kw)
    stages = abi.STAGE_ALL
    ow, oh = wl.out_size(stages)
    exp, _ = pyoracle.vardct_render(wl.desc(), stages, ow, oh)
    transport = str(rng.choice(["grouped", "grouped", "dense_i32", "sparse_i16"]))
    shifts = None
    if transport == "grouped" and rng.random() < 0.3:   # a progressive frame: two or three passes
        shifts = [int(rng.integers(1, 5)), 0] if rng.random() < 0.5 else [int(rng.integers(3, 6)), int(rng.integers(1, 3)), 0]
    partial = None
    if transport == "grouped" and rng.random() < 0.3:
        # a truncated stream (allow_partial): some pass groups end early — with several passes every (pass, group) on its own
        import copy
        n_groups = ((w + 255) // 256) * ((h + 255) // 256)
        cut = {int(g): int(rng.integers(0, 200)) for g in rng.choice(n_groups, size=min(n_groups, int(rng.integers(1, 4))), replace=False)}
        wl_t = copy.copy(wl)
        if shifts:
            partial = {(int(rng.integers(0, len(shifts))), g): k for g, k in cut.items()}
            wl_t.coeff = wl.progressive_truncated_coeff(shifts, partial)
        else:
            partial = cut
            wl_t.coeff = wl.truncated_coeff(partial)
        exp, _ = pyoracle.vardct_render(wl_t.desc(), stages, ow, oh)
    dkw = dict(coeff_transport=transport)
    if shifts:
        dkw["pass_shifts"] = shifts
    if partial is not None:
        dkw["partial"] = partial
    f = ctx.vardct_upload(wl.desc(**dkw))
    try:
        got = ctx.vardct_render(f, stages)
        ok = same_bits(got, exp)
        rx, ry = int(rng.integers(0, ow)), int(rng.integers(0, oh))
        rw, rh = int(rng.integers(1, ow - rx + 1)), int(rng.integers(1, oh - ry + 1))
        reg = ctx.vardct_render_region(f, stages, (rx, ry, rw, rh))
        ok_r = same_bits(reg, exp[:, ry:ry + rh, rx:rx + rw])
    finally:
        f.free()
    return ok and ok_r, ("vardct", w, h, dict(kw, transport=transport, passes=shifts, partial=partial, region=(rx, ry, rw, rh), full_ok=bool(ok)))


def run_batch(ctx, rng):
    """A batched render of two to six frames of unrelated sizes and filter settings (the streaming post kernel, its
    border-ring launches and the list-fed transform families across frame boundaries), twice back to back."""
    n = int(rng.integers(2, 7))
    wls, params = [], []
    for _ in range(n):
        w, h = int(rng.integers(8, 900)), int(rng.integers(8, 600))
        kw = dict(seed=int(rng.integers(0, 1000)), epf_iters=int(rng.integers(0, 4)), gabor=bool(rng.integers(0, 2)),
                  nz_fraction=float(rng.choice([0.02, 0.15, 0.5])))
        wls.append(VardctWorkload(w, h, **kw))
        params.append((w, h, kw))
    exp = [pyoracle.vardct_render(wl.desc(), abi.STAGE_ALL, wl.width, wl.height)[0] for wl in wls]
    frames = [ctx.vardct_upload(wl.desc(coeff_transport="grouped")) for wl in wls]
    try:
        for _ in range(2):
            ctx.vardct_render_batch(frames, abi.STAGE_ALL)
        ctx.synchronize()
        oks = [same_bits(ctx.download_result(f), e) for f, e in zip(frames, exp)]
    finally:
        for f in frames:
            f.free()
    return all(oks), ("batch", n, 0, dict(frames=params, ok=oks))


def run_extra(ctx, rng):
    """An extra channel of random size, bit depth, sample type and upsampling shift on a small rendered frame."""
    wl = VardctWorkload(64, 40, seed=int(rng.integers(0, 1000)))
    log2 = int(rng.choice([0, 0, 1, 2, 3, 4]))
    w, h = int(rng.integers(2, 200 >> min(log2, 3))) + 1, int(rng.integers(2, 120 >> min(log2, 3))) + 1
    kw = dict(seed=int(rng.integers(0, 1000)), i16=bool(rng.integers(0, 2)), upsampling_log2=log2)
    if rng.random() < 0.25:
        kw.update(i16=False, bit_depth=32, float_sample=True, exp_bits=8) if rng.random() < 0.5 else kw.update(i16=True, bit_depth=16, float_sample=True, exp_bits=5)
    else:
        kw["bit_depth"] = int(rng.integers(1, 15 if kw["i16"] else 31))
    ec, keep = make_extra_channel(w, h, **kw)
    exp = pyoracle.extra_channel(ec)
    f = ctx.vardct_upload(wl.desc())
    try:
        ctx.vardct_render(f, abi.STAGE_ALL, to_host=False)
        got = ctx.render_extra(f, int(rng.integers(0, abi.MAX_EXTRA)), ec)
    finally:
        f.free()
    return bool(np.array_equal(got.view(np.uint32), exp.view(np.uint32))), ("extra", w, h, kw)


def main():
    seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else int(time.time()) & 0xFFFF
    rng = np.random.default_rng(seed)
    print("CANARY", runtime.gpu_canary())
    ctx = runtime.Context(0)
    t_end = time.time() + seconds
    n, bad = {"modular": 0, "vardct": 0, "batch": 0, "extra": 0}, []
    while time.time() < t_end:
        r = rng.random()
        fn = run_modular if r < 0.5 else (run_vardct if r < 0.8 else (run_batch if r < 0.92 else run_extra))
        try:
            ok, what = fn(ctx, rng)
        except runtime.JxlGpuError as e:
            # descriptors the library refuses are fine (the synthetic generator may draw them); anything else is a finding
            if e.code == abi.ERR_UNSUPPORTED:
                continue
            ok, what = False, ("error", str(e))
        n[what[0]] = n.get(what[0], 0) + 1
        if not ok:
            bad.append(what)
            print("MISMATCH", what, flush=True)
    ctx.close()
    print(f"fuzz seed {seed}: {n} cases, {len(bad)} mismatches", flush=True)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
