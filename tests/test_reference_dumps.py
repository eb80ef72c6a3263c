"""Replay of per-stage dump directories (jxl_oxide_amd/refdump.py; tools/ref_dump.md tells a jxl-oxide maintainer
where to write them from).  Every directory under $JXL_REF_DUMPS — dumps of a REAL jxl-oxide run, produced off
this image — is rebuilt into a descriptor and compared stage by stage with the oracle (CPU) and with the HIP path
(GPU).  That is what turns DESIGN.md's "parity: partial (oracle pinned to tables and unit vectors only)" into
"pinned to reference outputs"; none exists in the container (no rustc), so here the same code path runs on a
self-made dump: a synthetic frame saved with the oracle's stage outputs and loaded back.

Tolerance: 0 ULP when the reference ran its generic implementation (tools/ref_dump.md forces it); a dump made
with the SSE / AVX2 / NEON implementations is expected within the north-star 1 ULP... of what THEY compute, which
the reference itself only promises to a tolerance — set JXL_REF_DUMPS_MAX_ULP for such dumps."""
import os

import numpy as np
import pytest

from jxl_oxide_amd import abi, refdump
from jxl_oxide_amd.synth import VardctWorkload
from util import assert_ulp

MAX_ULP = int(os.environ.get("JXL_REF_DUMPS_MAX_ULP", "0"))


def _dump_dirs(tmp_path_factory, oracle):
    dirs = []
    root = os.environ.get("JXL_REF_DUMPS")
    if root and os.path.isdir(root):
        dirs += sorted(os.path.join(root, d) for d in os.listdir(root) if os.path.exists(os.path.join(root, d, "meta.json")))
    # the self-made dumps: the default pipeline, and upsampling + HDR colour
    for i, kw in enumerate((dict(), dict(epf_iters=3, upsampling=2, hdr_pq=True, intensity_target=4000.0))):
        p = str(tmp_path_factory.mktemp(f"selfdump{i}"))
        refdump.save(VardctWorkload(264, 200, seed=40 + i, **kw), p, oracle)
        if i == 0:   # an RGBA image: alpha at half resolution, and a 16-bit float depth channel at full resolution
            from jxl_oxide_amd.synth import make_extra_channel
            for idx, ekw in ((0, dict(w=132, h=100, bit_depth=8, upsampling_log2=1)),
                             (1, dict(w=264, h=200, bit_depth=16, float_sample=True, exp_bits=5))):
                e = dict(ekw)
                ec, keep = make_extra_channel(e.pop("w"), e.pop("h"), seed=idx, **e)
                refdump.save_extra_channel(p, idx, ec, keep[0], oracle)
        dirs.append(p)
    return dirs


@pytest.fixture(scope="module")
def dumps(tmp_path_factory, oracle):
    return [refdump.load(p) for p in _dump_dirs(tmp_path_factory, oracle)]


def _render(oracle, dump, d, name):
    if name == "lf":
        _, lf = oracle.vardct_render(d, abi.STAGE_LF, dump.width, dump.height, want_lf=True, w8=dump.w8, h8=dump.h8)
        return lf
    ow, oh = dump.out_size() if name == "out" else (dump.width, dump.height)
    return oracle.vardct_render(d, refdump.STAGES[name], ow, oh)[0]


def test_oracle_matches_every_dumped_stage(oracle, dumps):
    assert dumps
    for dump in dumps:
        d = dump.desc()
        checked = 0
        for name in ("lf", "transform", "filters", "out"):
            ref = dump.stage(name)
            if ref is None:
                continue
            assert_ulp(_render(oracle, dump, d, name), ref, MAX_ULP, f"{dump.path}: oracle vs dumped stage '{name}'")
            checked += 1
        assert checked, f"{dump.path}: no stage files"


def test_oracle_matches_every_dumped_extra_channel(oracle, dumps):
    seen = 0
    for dump in dumps:
        for i, ec, exp in dump.extra_channels():
            if exp is None:
                continue
            assert_ulp(oracle.extra_channel(ec)[None], exp[None], MAX_ULP, f"{dump.path}: oracle vs dumped extra channel {i}")
            seen += 1
    assert seen >= 2   # the self-made RGBA dump


def test_loaded_descriptor_equals_the_one_it_was_saved_from(oracle, tmp_path):
    """Round trip of the format itself: the descriptor rebuilt from the files renders the same bits as the
    generator's own descriptor (every field that matters went through meta.json / the .npy files)."""
    wl = VardctWorkload(300, 264, seed=5, epf_iters=1)
    refdump.save(wl, str(tmp_path), None)
    dump = refdump.load(str(tmp_path))
    a, _ = oracle.vardct_render(wl.desc(), abi.STAGE_ALL, 300, 264)
    b, _ = oracle.vardct_render(dump.desc(), abi.STAGE_ALL, 300, 264)
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))


@pytest.mark.gpu
def test_device_matches_every_dumped_stage(gpu_ctx, oracle, dumps):
    for dump in dumps:
        d = dump.desc()
        f = gpu_ctx.vardct_upload(d)
        try:
            for name in ("lf", "transform", "filters", "out"):
                ref = dump.stage(name)
                if ref is None:
                    continue
                if name == "lf":
                    gpu_ctx.vardct_render(f, abi.STAGE_LF, to_host=False)
                    got = gpu_ctx.download_lf(f, dump.w8, dump.h8)
                else:
                    got = gpu_ctx.vardct_render(f, refdump.STAGES[name])
                assert_ulp(got, ref, max(MAX_ULP, 0), f"{dump.path}: device vs dumped stage '{name}'")
        finally:
            f.free()
