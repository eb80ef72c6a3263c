#!/usr/bin/env python3
"""PCIe-inclusive cost of handing one 4K frame's decoded state to the device, per coefficient
transport, and of the one-shot host-to-host call.  Never the headline number
(bench.py times HBM-resident frames); DESIGN.md §4 quotes these figures.

    python tools/bench_host.py [--width 3840 --height 2160 --reps 5]
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--width", type=int, default=3840)
    ap.add_argument("--height", type=int, default=2160)
    ap.add_argument("--reps", type=int, default=5)
    args = ap.parse_args()

    import numpy as np
    from jxl_oxide_amd import abi, runtime
    from jxl_oxide_amd.synth import VardctWorkload

    runtime.gpu_canary()
    ctx = runtime.Context(0)
    wl = VardctWorkload(args.width, args.height, seed=2000)
    mp = args.width * args.height / 1e6
    nz = float(np.count_nonzero(wl.coeff)) / wl.coeff.size
    out = {"workload": f"{args.width}x{args.height} VarDCT d1", "nonzero_fraction": round(nz, 4), "transports": {}}
    ref = None
    for tr in ("dense_i32", "dense_i16", "sparse_i32", "sparse_i16"):
        d = wl.desc(coeff_transport=tr)
        if tr.startswith("dense"):
            h2d = 3 * wl.wr * wl.hr * (2 if tr.endswith("i16") else 4)
        else:
            h2d = sum(int(d.sparse_count[c]) for c in range(3)) * (4 + (2 if tr.endswith("i16") else 4))
        best_up, best_all = 1e9, 1e9
        for _ in range(args.reps):
            t0 = time.perf_counter()
            f = ctx.vardct_upload(d)
            t1 = time.perf_counter()
            ctx.vardct_render(f, abi.STAGE_ALL, to_host=False)
            ctx.synchronize()
            t2 = time.perf_counter()
            best_up, best_all = min(best_up, t1 - t0), min(best_all, t2 - t0)
            if ref is None:
                ref = ctx.vardct_render(f, abi.STAGE_ALL)
            elif _ == 0:
                got = ctx.vardct_render(f, abi.STAGE_ALL)
                assert np.array_equal(got.view(np.uint32), ref.view(np.uint32)), tr
            f.free()
        out["transports"][tr] = {
            "coeff_h2d_MB": round(h2d / 1e6, 1), "upload_ms": round(best_up * 1e3, 2),
            "upload_plus_render_ms": round(best_all * 1e3, 2), "MP_per_s_upload_plus_render": round(mp / best_all, 1),
        }
    buf = np.zeros((3, args.height, args.width), dtype=np.float32)
    for tr in ("dense_i32", "sparse_i16"):
        d = wl.desc(coeff_transport=tr)
        best = 1e9
        for _ in range(args.reps + 1):
            t0 = time.perf_counter()
            ctx.vardct_render_host(d, abi.STAGE_ALL, args.width, args.height, out=buf)
            best = min(best, time.perf_counter() - t0)
        assert np.array_equal(buf.view(np.uint32), ref.view(np.uint32)), tr
        out[f"render_host_{tr}_ms"] = round(best * 1e3, 2)
        out[f"render_host_{tr}_MP_per_s"] = round(mp / best, 1)
    # u8 interleaved output formatted on the device: 3 B/px over PCIe instead of 12
    f = ctx.vardct_upload(wl.desc(coeff_transport="sparse_i16"))
    best = 1e9
    for _ in range(args.reps):
        t0 = time.perf_counter()
        ctx.vardct_render(f, abi.STAGE_ALL, to_host=False)
        ctx.format_output(f, abi.FMT_U8, 1)
        best = min(best, time.perf_counter() - t0)
    f.free()
    out["render_plus_u8_download_ms"] = round(best * 1e3, 2)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
