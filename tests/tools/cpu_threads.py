import os, subprocess, sys
code = r'''
import sys, time, os
sys.path.insert(0, ".")
import numpy as np
from jxl_oxide_amd.synth import VardctWorkload
from oracle import pyoracle
wl = VardctWorkload(3840, 2160, seed=2000)
d = wl.desc(); buf = np.zeros((3, 2160, 3840), np.float32)
pyoracle.vardct_render(d, 63, 3840, 2160, out=buf)
t = time.time(); n = 0
while time.time() - t < 4: pyoracle.vardct_render(d, 63, 3840, 2160, out=buf); n += 1
print(os.environ["OMP_NUM_THREADS"], round(n * 8.2944 / (time.time() - t), 1), "MP/s")
'''
for th in sys.argv[1:]:
    env = dict(os.environ, OMP_NUM_THREADS=th, OMP_PROC_BIND="close", OMP_WAIT_POLICY="active")
    print(subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True).stdout.strip(), flush=True)
