#!/usr/bin/env python3
"""Numbers for DESIGN §8.1 ("one pass"): what a fused transform + post workgroup would cost on the §8(d) shape mix.

A workgroup owns a rectangle of OUTPUT pixels (band height x piece width), needs the transform output of that rectangle
plus the post stage's halo (Gabor 1 + EPF steps 1 and 2: 3 + 3 → 8 rows / columns rounded to whole cells), and therefore has
to run every varblock that intersects the padded rectangle.  This script draws the synthetic 4K block map bench.py uses
(jxl_oxide_amd.synth.draw_tiling, the mix of SURVEY §8(d)) and reports, per tile shape:
  dup      transform work (coefficient samples pushed through the inverse DCT) relative to one pass over the frame
  lds_kb   LDS the tile needs for three f32 planes of the padded rectangle (+ whole varblocks that stick out are NOT kept:
           only the rows / columns inside the padded rectangle are stored)
  wg       workgroups per 4K frame; hbm_mb: list words in (15 % non-zeros, re-read dup times) + planes out
CPU only; no product code is involved.  `python tools/one_pass_model.py`"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from jxl_oxide_amd import abi  # noqa: E402
from jxl_oxide_amd.synth import DCT_SELECT_SIZE, SEED_BASE, draw_tiling  # noqa: E402

W, H = 3840, 2160
HALO = 8          # px: rows / columns of transform output the post stage reads beyond its own (Gabor + EPF 1 + EPF 2)
NZ = 0.15         # fraction of non-zero coefficients


def main():
    rng = np.random.default_rng(SEED_BASE + 2)
    w8, h8 = W // 8, H // 8
    kind, _ = draw_tiling(rng, w8, h8)
    ys, xs = np.nonzero(kind <= 26)
    t = kind[ys, xs]
    bw = np.array([DCT_SELECT_SIZE[int(k)][0] for k in t]) * 8
    bh = np.array([DCT_SELECT_SIZE[int(k)][1] for k in t]) * 8
    x0, y0 = xs * 8, ys * 8
    area = (bw * bh).astype(np.int64)
    total = int(area.sum())
    assert total == W * H, (total, W * H)
    big = np.maximum(bw, bh)
    print(f"4K block map: {len(t)} varblocks; share of the area by longest side: " +
          ", ".join(f"{s}px {area[big == s].sum() / total:.3f}" for s in (8, 16, 32, 64)))
    print(f"{'tile (h x w)':>14s} {'dup':>6s} {'lds_kb':>7s} {'wg':>6s} {'hbm_mb':>7s}   (halo {HALO} px)")
    out_mb = W * H * 12 / 1e6
    list_mb = W * H * 3 * NZ * 4 / 1e6          # one 32-bit list word per non-zero coefficient
    for th, tw in ((64, 128), (64, 256), (128, 128), (128, 256), (256, 256), (256, 128), (32, 256)):
        work = 0
        for ty in range(0, H, th):
            ry0, ry1 = max(ty - HALO, 0), min(ty + th + HALO, H)
            in_y = (y0 < ry1) & (y0 + bh > ry0)
            for tx in range(0, W, tw):
                rx0, rx1 = max(tx - HALO, 0), min(tx + tw + HALO, W)
                sel = in_y & (x0 < rx1) & (x0 + bw > rx0)
                work += int(area[sel].sum())
        dup = work / total
        lds = 3 * (th + 2 * HALO) * (tw + 2 * HALO) * 4 / 1024
        wg = -(-H // th) * -(-W // tw)
        print(f"{th:>6d} x {tw:<5d} {dup:6.2f} {lds:7.1f} {wg:6d} {list_mb * dup + out_mb + 9.3:7.1f}")
    # a persistent workgroup walking DOWN a column strip with a rolling window of rows in LDS: every varblock is transformed once
    # per strip it touches (no vertical duplication); the window holds the tallest varblock (64 rows) + the halo rows on both sides
    print(f"{'strip width':>14s} {'dup':>6s} {'lds_kb':>7s} {'wg':>6s} {'hbm_mb':>7s}   (rolling window of {64 + 2 * HALO} rows)")
    for tw in (64, 120, 128, 256, 512):
        work = 0
        for tx in range(0, W, tw):
            rx0, rx1 = max(tx - HALO, 0), min(tx + tw + HALO, W)
            sel = (x0 < rx1) & (x0 + bw > rx0)
            work += int(area[sel].sum())
        dup = work / total
        lds = 3 * (64 + 2 * HALO) * (tw + 2 * HALO) * 4 / 1024
        print(f"{tw:>14d} {dup:6.2f} {lds:7.1f} {-(-W // tw):6d} {list_mb * dup + out_mb + 9.3:7.1f}")
    print(f"two-pass design as measured (profiles/r05_pmc_hbm_traffic.json): 355.5 MB per frame; one-pass algorithmic (SURVEY §8(d)): 201.8 MB")


if __name__ == "__main__":
    main()
