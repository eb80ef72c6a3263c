#!/usr/bin/env python3
"""GPU-vs-GPU check of the env-selected streaming variants on one frame (no oracle, no torch):
the default kernels are the reference.  One subprocess per variant (switches are read at load)."""
import hashlib
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, hashlib
sys.path.insert(0, %r)
from jxl_oxide_amd import abi, runtime
from jxl_oxide_amd.synth import VardctWorkload
wl = VardctWorkload(520, 264, seed=1)
ctx = runtime.Context(0)
f = ctx.vardct_upload(wl.desc())
out = ctx.vardct_render(f, abi.STAGE_ALL)
print(hashlib.sha256(out.tobytes()).hexdigest())
''' % ROOT

if __name__ == "__main__":
    variants = sys.argv[1:] or ["", "JXLGPU_STREAM_SPLIT=1", "JXLGPU_STREAM_SPLIT=2", "JXLGPU_STREAM_SPLIT=3", "JXLGPU_STREAM_PK=3"]
    ref = None
    for v in variants:
        env = dict(os.environ)
        for kv in v.split():
            k, _, val = kv.partition("=")
            env[k] = val
        r = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True, timeout=60)
        h = r.stdout.strip().splitlines()[-1] if r.stdout.strip() else "FAILED " + r.stderr[-200:]
        ref = ref or h
        print(f"{v or '(default)':40s} {'same' if h == ref else 'DIFFERENT'}  {h[:16]}", flush=True)
