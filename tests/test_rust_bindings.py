"""bindings/jxlgpu.rs — the Rust side of the C ABI.  No rustc in the image, so the file is
generated from include/jxlgpu.h (tools/gen_rust_bindings.py) and pinned three ways: the committed file is the
generator's output for the current header; the repr(C) size the generator derives for every struct equals the C
compiler's (through the ctypes mirror, whose offsets tests/test_abi.py checks against gcc); the extern block
declares exactly the functions the shared library exports."""
import ctypes as C
import importlib.util
import os
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _gen():
    spec = importlib.util.spec_from_file_location("gen_rust_bindings", os.path.join(ROOT, "tools", "gen_rust_bindings.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_committed_bindings_match_the_header():
    text, _, _ = _gen().generate()
    assert open(os.path.join(ROOT, "bindings", "jxlgpu.rs")).read() == text, "run python tools/gen_rust_bindings.py"


def test_struct_sizes_equal_the_c_layout():
    from jxl_oxide_amd import abi
    _, sizes, _ = _gen().generate()
    mirror = {
        "JxlGpuFilterParams": abi.FilterParams, "JxlGpuColorParams": abi.ColorParams, "JxlGpuBlendRect": abi.BlendRect,
        "JxlGpuNoiseParams": abi.NoiseParams, "JxlGpuUpsampling": abi.Upsampling, "JxlGpuLfGroup": abi.LfGroup,
        "JxlGpuHfGroup": abi.HfGroup, "JxlGpuVardctDesc": abi.VardctDesc, "JxlGpuRegion": abi.Region, "JxlGpuOut": abi.Out,
        "JxlGpuFormatDesc": abi.FormatDesc, "JxlGpuExtraChannel": abi.ExtraChannel, "JxlGpuSqueezeStep": abi.SqueezeStep, "JxlGpuTransform": abi.Transform,
        "JxlGpuModularChannel": abi.ModularChannel, "JxlGpuModularDesc": abi.ModularDesc, "JxlGpuMaLeaf": abi.MaLeaf,
    }
    assert set(mirror) == set(sizes), set(mirror) ^ set(sizes)
    for name, cls in mirror.items():
        assert sizes[name][0] == C.sizeof(cls), (name, sizes[name], C.sizeof(cls))
        assert sizes[name][1] == C.alignment(cls), (name, sizes[name], C.alignment(cls))


def test_extern_block_declares_what_the_library_exports():
    _, _, funcs = _gen().generate()
    lib = os.path.join(ROOT, "jxl-oxide_amd", "csrc", "libjxlgpu.so")
    out = subprocess.check_output(["nm", "-D", "--defined-only", lib], text=True)
    exported = set(re.findall(r"\sT\s+(jxlgpu_\w+)", out))
    assert exported == set(funcs), exported ^ set(funcs)
    rs = open(os.path.join(ROOT, "bindings", "jxlgpu.rs")).read()
    for f in funcs:
        assert f"pub fn {f}(" in rs
