"""Extracts the default upsampling weight tables D_UP2/4/8 (pure data,
jxl-image/src/lib.rs:533-) from the reference checkout into an .npz.
Run in the build container only (the reference is not present on the GPU box)."""
import re
import sys

import numpy as np

src = open("/root/reference/crates/jxl-image/src/lib.rs").read()
out = {}
for name, n in (("D_UP2", 15), ("D_UP4", 55), ("D_UP8", 210)):
    i = src.index(f"const {name}:")
    body = src[i:src.index("];", i)]
    body = body[body.index("= [") + 3:]
    vals = [float(v) for v in re.findall(r"-?\d+\.\d+", body)]
    assert len(vals) == n, (name, len(vals))
    out[name] = np.array(vals, dtype=np.float32)
np.savez(sys.argv[1], up2=out["D_UP2"], up4=out["D_UP4"], up8=out["D_UP8"])
