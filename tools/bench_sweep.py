#!/usr/bin/env python3
"""A/B the environment-selected kernel variants in one GPU call: each variant runs bench.py in its
own process (the switches are read once, at library load) and one table comes out.

    python tools/bench_sweep.py                       # the built-in list
    python tools/bench_sweep.py "JXLGPU_STREAM_PK=1 JXLGPU_STREAM_ROWS=56" "JXLGPU_NO_DEQ_LUT=1"

Columns: whole-job GP/s (4 contexts, overlapped), ms per 8-frame step, and the isolated launch
times of the transform and post groups (one stream busy).  ~9 s per variant.
"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEFAULT = ["", "JXLGPU_STREAM_SPLIT=1", "JXLGPU_STREAM_SPLIT=2", "JXLGPU_STREAM_SPLIT=3", "JXLGPU_STREAM_PK=1", "JXLGPU_STREAM_PK=3", "JXLGPU_STREAM_PK=3 JXLGPU_STREAM_ROWS=36",
           "JXLGPU_STREAM_PK=1 JXLGPU_STREAM_ROWS=36", "JXLGPU_NO_DEQ_LUT=1", "JXLGPU_STREAM_ROWS=40", "JXLGPU_STREAM_ROWS=56"]


def main():
    variants = sys.argv[1:] or DEFAULT
    steps = os.environ.get("SWEEP_STEPS", "30")
    print(f"{'variant':60s} {'GP/s':>8s} {'ms/step':>8s} {'transform ms':>13s} {'post ms':>8s}")
    for v in variants:
        env = dict(os.environ)
        for kv in v.split():
            k, _, val = kv.partition("=")
            env[k] = val
        try:
            out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", "--steps", steps],
                                 env=env, capture_output=True, text=True, timeout=180).stdout.strip().splitlines()[-1]
            d = json.loads(out)
            g = d["roofline"]["other_group_ms"]
            print(f"{v or '(default)':60s} {d['value'] / 1e3:8.2f} {d['ms_per_step']:8.4f} {g['transform']:13.4f} {g['post']:8.4f}", flush=True)
        except Exception as e:  # keep going: one broken variant must not cost the whole GPU call
            print(f"{v or '(default)':60s} failed: {e}", flush=True)


if __name__ == "__main__":
    main()
