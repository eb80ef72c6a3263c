"""The -O3 -march=native build of the oracle (bench.py's cpu_baseline timing copy) must produce
the same bits as the -O2 build the parity tests use: same sources, -ffp-contract=off in both."""
import numpy as np

from jxl_oxide_amd import abi
from jxl_oxide_amd.synth import VardctWorkload
from jxl_oxide_amd.synth_modular import ModularWorkload


def test_native_build_is_bit_identical(oracle):
    wl = VardctWorkload(300, 264, seed=21)
    ml = ModularWorkload(200, 136, kind="squeeze", lossy=True, epf_iters=1, seed=2)
    stages_m = abi.STAGE_ALL | abi.STAGE_MODULAR_TO_FLOAT
    try:
        ref_v, _ = oracle.vardct_render(wl.desc(), abi.STAGE_ALL, wl.width, wl.height)
        ref_m = oracle.modular_render(ml.desc(), stages_m, ml.width, ml.height)
        oracle.use_native(True)
        got_v, _ = oracle.vardct_render(wl.desc(), abi.STAGE_ALL, wl.width, wl.height)
        got_m = oracle.modular_render(ml.desc(), stages_m, ml.width, ml.height)
    finally:
        oracle.use_native(False)
    assert np.array_equal(ref_v.view(np.uint32), got_v.view(np.uint32))
    assert np.array_equal(ref_m.view(np.uint32), got_m.view(np.uint32))
