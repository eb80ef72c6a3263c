"""csrc/fast_math_device.h on the device: the residual-chain square root and the shared-reciprocal quotient the HDR colour
chain of the 2x upsampling kernel uses instead of the compiler's IEEE expansions must give the SAME BITS on their stated
ranges — every float of the square root's range, 2^31 random pairs for the quotient (tests/c/fast_math_check.hip; v_rsq_f32 /
v_rcp_f32 cannot be emulated on the host, so this runs where the kernels run).  The end-to-end gate is the config-5 parity
test against the oracle (tests/test_gpu_baseline_sizes.py, tests/test_gpu_vardct.py::test_hdr_pq_chain and the dark-image case)."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_fast_sqrt_and_division_equal_the_ieee_expansions_on_the_device():
    exe = os.path.join(ROOT, "tools", "_bin", "fast_math_check")
    if not os.path.exists(exe):
        subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-ffp-contract=off", "-fhip-fp32-correctly-rounded-divide-sqrt",
                               "-fno-gpu-flush-denormals-to-zero", "-Wno-unused-result", "-I", os.path.join(ROOT, "jxl-oxide_amd", "csrc"),
                               os.path.join(ROOT, "tests", "c", "fast_math_check.hip"), "-o", exe])
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    print(r.stdout)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.count("mismatches 0") == 4, r.stdout
