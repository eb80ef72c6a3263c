"""jxl-oxide_amd — MI355X-native JPEG XL transform-and-render hot path.

The product is csrc/libjxlgpu.so (hand-written HIP for gfx950 behind the C ABI of
include/jxlgpu.h).  This Python package is plumbing: ctypes bindings (abi, runtime), the synthetic
boundary-state generator (synth), the default dequant tables (dequant) and the multi-GPU sharding
helper (shard)."""
from . import abi  # noqa: F401
