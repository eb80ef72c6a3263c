#!/usr/bin/env python3
"""Regenerates tests/golden/oracle_digests.json: SHA-256 of the oracle's output on a fixed set of
seeded synthetic workloads.  The upstream golden vectors (libjxl conformance suite) are not in the
container and the reference cannot be built here, so these digests do NOT pin the oracle to the
reference; they pin it to itself across refactors — any arithmetic change in oracle/ shows up as
a digest mismatch in tests/test_oracle_digests.py and has to be justified.

    python tests/golden/make_oracle_digests.py
"""
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault("OMP_NUM_THREADS", "4")


def cases():
    from jxl_oxide_amd import abi
    from jxl_oxide_amd.synth import JpegWorkload, VardctWorkload
    from jxl_oxide_amd.synth_modular import ModularWorkload
    from oracle import pyoracle
    S = abi.STAGE_ALL

    def vardct(**kw):
        wl = VardctWorkload(**kw)
        w, h = wl.out_size(S)
        return lambda: pyoracle.vardct_render(wl.desc(), S, w, h)[0]

    def jpeg(**kw):
        wl = JpegWorkload(**kw)
        return lambda: pyoracle.vardct_render(wl.desc(), S, wl.width, wl.height)[0]

    def modular(**kw):
        wl = ModularWorkload(**kw)
        return lambda: wl_planes(pyoracle.modular_inverse(wl.desc(), wl.shapes(), wl.dtype))

    def truncated():
        import copy
        wl = VardctWorkload(width=600, height=520, seed=12, nz_fraction=0.2)
        shifts = [3, 1, 0]
        partial = {(0, 4): 61, (1, 1): 40, (1, 7): 0, (2, 0): 0, (2, 1): 17, (2, 4): 5}
        wl_sum = copy.copy(wl)
        wl_sum.coeff = wl.progressive_truncated_coeff(shifts, partial)
        return lambda: pyoracle.vardct_render(wl_sum.desc(), S, 600, 520)[0]

    def wl_planes(planes):
        import numpy as np
        return np.concatenate([np.ascontiguousarray(p).reshape(-1).astype(np.int64) for p in planes])

    return {
        "vardct_default_264x200": vardct(width=264, height=200, seed=1),
        "vardct_epf3_pq_200x136": vardct(width=200, height=136, seed=2, epf_iters=3, intensity_target=4000.0, hdr_pq=True),
        "vardct_up2_noise_72x56": vardct(width=72, height=56, seed=3, upsampling=2, epf_iters=1, noise=True),
        "vardct_tonemap_200x136": vardct(width=200, height=136, seed=4, epf_iters=1, intensity_target=4000.0, color_mode="tone_map_srgb"),
        "vardct_bigblocks_520x264": vardct(width=520, height=264, seed=5, lf_i16=False),
        "jpeg_420_83x45": jpeg(width=83, height=45, mode="420", seed=6),
        "jpeg_mixed_300x270": jpeg(width=300, height=270, mode="mixed", seed=7),
        "modular_squeeze_i16_200x136": modular(width=200, height=136, kind="squeeze", lossy=True, i16=True, seed=8),
        "modular_predictor6_70x33": modular(width=70, height=33, kind="predictor", predictor=6, i16=False, seed=9),
        "modular_palette_delta_37x21": modular(width=37, height=21, kind="palette_delta", predictor=5, i16=True, seed=10),
        # round 5: group_dim 1024 (subgrids wider than 512 columns; Squeeze sub-channels carved on the 1024 grid) and the sum a
        # truncated progressive stream leaves (allow_partial with several passes)
        "modular_squeeze_wp_gd1024_1100x600": modular(width=1100, height=600, kind="squeeze", lossy=False, xyb=False, residual=6, seed=11, group_dim=1024),
        "vardct_truncated_progressive_600x520": truncated(),
    }


def digests():
    import numpy as np
    out = {}
    for name, fn in cases().items():
        a = np.ascontiguousarray(fn())
        out[name] = hashlib.sha256(a.tobytes()).hexdigest()
    return out


if __name__ == "__main__":
    d = digests()
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "oracle_digests.json")
    json.dump(d, open(path, "w"), indent=1, sort_keys=True)
    print(f"wrote {len(d)} digests to {path}")
