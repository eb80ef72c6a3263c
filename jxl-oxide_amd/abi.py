"""ctypes mirror of include/jxlgpu.h and the loader for libjxlgpu.so.

This is plumbing for tests/bench: the product is the shared library.  The loader fails loudly if
the HIP library is missing — there is no CPU fallback in this package.
"""
import ctypes as C
import os

ABI_VERSION = 24
NUM_TRANSFORMS = 27

OK = 0
ERR_INVALID_ARG = -1
ERR_OOM = -2
ERR_DEVICE = -3
ERR_UNSUPPORTED = -4
ERR_ABI = -5

BLOCK_OCCUPIED = 0xFE
BLOCK_UNINIT = 0xFF
SAMPLE_I32 = 0
SAMPLE_I16 = 1

TF_LINEAR, TF_SRGB, TF_PQ, TF_BT709, TF_GAMMA, TF_HLG = range(6)
GAMUT_NONE, GAMUT_MAP, GAMUT_CLIP = range(3)
COEFF_DENSE, COEFF_SPARSE, COEFF_GROUPED = range(3)

STAGE_LF = 0x01
STAGE_TRANSFORM = 0x02
STAGE_GABOR = 0x04
STAGE_EPF = 0x08
STAGE_UPSAMPLE = 0x10
STAGE_COLOR = 0x20
STAGE_NOISE = 0x80
STAGE_ALL = 0xBF
STAGE_MODULAR_INVERSE = 0x02
STAGE_MODULAR_TO_FLOAT = 0x40

FMT_F32, FMT_U16, FMT_U8 = 0, 1, 2

MEM_HOST = 0
MEM_DEVICE = 1
MEM_HOST_PINNED = 2

TR_RCT, TR_PALETTE, TR_SQUEEZE = 0, 1, 2
LEAF_BY_ROW, LEAF_BY_COLUMN = 14, 15   # JxlGpuMaLeaf.predictor in unit_leaves: the unit's leaves come per row / column from axis_leaves

# TransformType (jxl-vardct/src/dct_select.rs:4-32) -> (bw, bh) in 8x8 cells (:52-76)
DCT_SELECT_SIZE = [
    (1, 1), (1, 1), (1, 1), (1, 1), (2, 2), (4, 4), (1, 2), (2, 1), (1, 4),
    (4, 1), (2, 4), (4, 2), (1, 1), (1, 1), (1, 1), (1, 1), (1, 1), (1, 1),
    (8, 8), (4, 8), (8, 4), (16, 16), (8, 16), (16, 8), (32, 32), (16, 32), (32, 16),
]
TRANSFORM_NAMES = [
    "Dct8", "Hornuss", "Dct2", "Dct4", "Dct16", "Dct32", "Dct16x8", "Dct8x16", "Dct32x8",
    "Dct8x32", "Dct32x16", "Dct16x32", "Dct4x8", "Dct8x4", "Afv0", "Afv1", "Afv2", "Afv3",
    "Dct64", "Dct64x32", "Dct32x64", "Dct128", "Dct128x64", "Dct64x128", "Dct256",
    "Dct256x128", "Dct128x256",
]

f32p = C.POINTER(C.c_float)
i32p = C.POINTER(C.c_int32)
u8p = C.POINTER(C.c_uint8)


class FilterParams(C.Structure):
    _fields_ = [
        ("gab_enabled", C.c_uint32),
        ("gab_weights", (C.c_float * 2) * 3),
        ("epf_iters", C.c_uint32),
        ("epf_channel_scale", C.c_float * 3),
        ("epf_pass0_sigma_scale", C.c_float),
        ("epf_pass2_sigma_scale", C.c_float),
        ("epf_border_sad_mul", C.c_float),
        ("epf_sigma_for_modular", C.c_float),
    ]


class ColorParams(C.Structure):
    _fields_ = [
        ("enabled", C.c_uint32),
        ("opsin_bias", C.c_float * 3),
        ("intensity_target", C.c_float),
        ("matrix", C.c_float * 9),
        ("gamut_map", C.c_uint32),
        ("gamut_luminances", C.c_float * 3),
        ("gamut_saturation_factor", C.c_float),
        ("has_matrix2", C.c_uint32),
        ("matrix2", C.c_float * 9),
        ("transfer_function", C.c_uint32),
        ("gamma", C.c_float),
        ("hlg_luminances", C.c_float * 3),
        ("tone_map", C.c_uint32),
        ("tm_luminances", C.c_float * 3),
        ("tm_min_nits", C.c_float),
        ("tm_target_display_luminance", C.c_float),
        ("tm_gamut_map", C.c_uint32),
        ("tm_gamut_saturation_factor", C.c_float),
        ("ycbcr", C.c_uint32),
        ("hlg_ootf_intensity_target", C.c_float),
    ]


(BLEND_REPLACE, BLEND_ADD, BLEND_MUL, BLEND_BLEND, BLEND_MULADD, BLEND_MIXALPHA, BLEND_SKIP) = range(7)


class BlendRect(C.Structure):
    _fields_ = [
        ("mode", C.c_uint32),
        ("clamp", C.c_uint32),
        ("swapped", C.c_uint32),
        ("premultiplied", C.c_uint32),
        ("base_alpha", C.c_void_p),
        ("base_alpha_stride", C.c_uint32),
        ("new_alpha", C.c_void_p),
        ("new_alpha_stride", C.c_uint32),
        ("base_x", C.c_uint32),
        ("base_y", C.c_uint32),
        ("new_x", C.c_uint32),
        ("new_y", C.c_uint32),
        ("width", C.c_uint32),
        ("height", C.c_uint32),
    ]


class NoiseParams(C.Structure):
    _fields_ = [
        ("enabled", C.c_uint32),
        ("lut", C.c_float * 8),
        ("visible_frames", C.c_uint32),
        ("invisible_frames", C.c_uint32),
    ]


class Upsampling(C.Structure):
    _fields_ = [
        ("factor", C.c_uint32),
        ("up2_weight", f32p),
        ("up4_weight", f32p),
        ("up8_weight", f32p),
    ]


class LfGroup(C.Structure):
    _fields_ = [
        ("width_px", C.c_uint32),
        ("height_px", C.c_uint32),
        ("lf_quant", C.c_void_p * 3),
        ("extra_precision", C.c_uint32),
        ("has_hf_meta", C.c_uint32),
        ("block_kind", u8p),
        ("hf_mul", i32p),
        ("epf_sigma", f32p),
        ("x_from_y", i32p),
        ("b_from_y", i32p),
    ]


class HfGroup(C.Structure):
    _fields_ = [
        ("num_varblocks", C.c_uint32),
        ("num_nz", C.c_uint32),
        ("nz_count", C.POINTER(C.c_uint16)),
        ("nz", C.POINTER(C.c_uint32)),
    ]


class VardctDesc(C.Structure):
    _fields_ = [
        ("abi", C.c_uint32),
        ("width", C.c_uint32),
        ("height", C.c_uint32),
        ("group_dim", C.c_uint32),
        ("lf_sample_type", C.c_uint32),
        ("jpeg_upsampling", C.c_uint32 * 3),
        ("coeff", C.c_void_p * 3),
        ("coeff_stride", C.c_uint32),
        ("coeff_format", C.c_uint32),
        ("coeff_sample_type", C.c_uint32),
        ("sparse_pos", C.POINTER(C.c_uint32) * 3),
        ("sparse_count", C.c_uint64 * 3),
        ("num_hf_groups", C.c_uint32),
        ("hf_groups", C.POINTER(HfGroup)),
        ("num_passes", C.c_uint32),
        ("allow_partial", C.c_uint32),
        ("lf_frame", f32p * 3),
        ("lf_frame_stride", C.c_uint32),
        ("num_lf_groups", C.c_uint32),
        ("lf_groups", C.POINTER(LfGroup)),
        ("global_scale", C.c_uint32),
        ("quant_lf", C.c_uint32),
        ("m_lf", C.c_float * 3),
        ("colour_factor", C.c_uint32),
        ("base_correlation_x", C.c_float),
        ("base_correlation_b", C.c_float),
        ("x_factor_lf", C.c_uint32),
        ("b_factor_lf", C.c_uint32),
        ("x_qm_scale", C.c_uint32),
        ("b_qm_scale", C.c_uint32),
        ("quant_bias", C.c_float * 3),
        ("quant_bias_numerator", C.c_float),
        ("skip_adaptive_lf_smoothing", C.c_uint32),
        ("dequant", (f32p * 3) * NUM_TRANSFORMS),
        ("sec_half_large", f32p * 3),
        ("filter", FilterParams),
        ("upsampling", Upsampling),
        ("noise", NoiseParams),
        ("color", ColorParams),
    ]


class Region(C.Structure):
    _fields_ = [("left", C.c_int32), ("top", C.c_int32), ("width", C.c_uint32), ("height", C.c_uint32)]


class Out(C.Structure):
    _fields_ = [("planes", f32p * 3), ("stride", C.c_uint32), ("mem", C.c_uint32)]


MAX_EXTRA = 8
IPC_HANDLE_BYTES = 64


class FormatDesc(C.Structure):
    _fields_ = [("sample_format", C.c_uint32), ("orientation", C.c_uint32),
                ("num_extra", C.c_uint32), ("extra", C.c_uint32 * 4)]


class ExtraChannel(C.Structure):
    _fields_ = [
        ("data", C.c_void_p),
        ("width", C.c_uint32), ("height", C.c_uint32),
        ("sample_type", C.c_uint32),
        ("bit_depth", C.c_uint32),
        ("float_sample", C.c_uint32),
        ("exp_bits", C.c_uint32),
        ("upsampling_log2", C.c_uint32),
        ("weights", Upsampling),
    ]


TRACE_FN = C.CFUNCTYPE(None, C.c_void_p, C.c_char_p, C.c_int)   # jxlgpu_trace_fn


class SqueezeStep(C.Structure):
    _fields_ = [("horizontal", C.c_uint32), ("in_place", C.c_uint32),
                ("begin_c", C.c_uint32), ("num_c", C.c_uint32)]


class Transform(C.Structure):
    _fields_ = [
        ("kind", C.c_uint32),
        ("begin_c", C.c_uint32),
        ("rct_type", C.c_uint32),
        ("num_c", C.c_uint32),
        ("nb_colours", C.c_uint32),
        ("nb_deltas", C.c_uint32),
        ("d_pred", C.c_uint32),
        ("wp_params", C.c_int32 * 11),
        ("num_sq", C.c_uint32),
        ("sq", C.POINTER(SqueezeStep)),
    ]


class ModularChannel(C.Structure):
    _fields_ = [("data", C.c_void_p), ("width", C.c_uint32), ("height", C.c_uint32)]


class MaLeaf(C.Structure):
    _fields_ = [("predictor", C.c_uint32), ("multiplier", C.c_int32), ("offset", C.c_int32)]


class ModularDesc(C.Structure):
    _fields_ = [
        ("abi", C.c_uint32),
        ("sample_type", C.c_uint32),
        ("bit_depth", C.c_uint32),
        ("num_channels", C.c_uint32),
        ("num_color_channels", C.c_uint32),
        ("channels", C.POINTER(ModularChannel)),
        ("num_meta_channels", C.c_uint32),
        ("meta_channels", C.POINTER(ModularChannel)),
        ("num_transforms", C.c_uint32),
        ("transforms", C.POINTER(Transform)),
        ("residual_predictor", C.c_uint32),
        ("residual_multiplier", C.c_int32),
        ("residual_offset", C.c_int32),
        ("wp_params", C.c_int32 * 11),
        ("group_dim", C.c_uint32),
        ("xyb_encoded", C.c_uint32),
        ("m_lf_unscaled", C.c_float * 3),
        ("float_sample", C.c_uint32),
        ("exp_bits", C.c_uint32),
        ("filter", FilterParams),
        ("upsampling", Upsampling),
        ("noise", NoiseParams),
        ("color", ColorParams),
        ("unit_leaves", C.POINTER(MaLeaf)),
        ("num_unit_leaves", C.c_uint32),
        ("axis_leaves", C.POINTER(MaLeaf)),
        ("num_axis_leaves", C.c_uint32),
    ]


# every symbol include/jxlgpu.h declares: (name, restype, argtypes)
_SYMBOLS = [
    ("jxlgpu_create", C.c_int, [C.c_int, C.POINTER(C.c_void_p)]),
    ("jxlgpu_destroy", None, [C.c_void_p]),
    ("jxlgpu_last_error", C.c_char_p, [C.c_void_p]),
    ("jxlgpu_abi_version", C.c_uint32, []),
    ("jxlgpu_synchronize", C.c_int, [C.c_void_p]),
    ("jxlgpu_stream", C.c_void_p, [C.c_void_p]),
    ("jxlgpu_frame_wait", C.c_int, [C.c_void_p, C.c_void_p]),
    ("jxlgpu_host_alloc", C.c_int, [C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p)]),
    ("jxlgpu_host_free", None, [C.c_void_p, C.c_void_p]),
    ("jxlgpu_upload_split", C.c_int, [C.c_void_p, C.POINTER(C.c_double)]),
    ("jxlgpu_selftest_libm", C.c_int, [C.c_void_p, C.c_int, f32p, C.c_size_t, C.c_float, f32p]),
    ("jxlgpu_set_memory_limit", C.c_int, [C.c_void_p, C.c_uint64]),
    ("jxlgpu_memory_usage", C.c_int, [C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    ("jxlgpu_set_trace", C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    ("jxlgpu_profile_select", C.c_int, [C.c_void_p, C.c_int]),
    ("jxlgpu_profile_read", C.c_int, [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_uint64)]),
    ("jxlgpu_vardct_upload", C.c_int, [C.c_void_p, C.POINTER(VardctDesc), C.POINTER(C.c_void_p)]),
    ("jxlgpu_vardct_render", C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, C.POINTER(Out)]),
    ("jxlgpu_vardct_render_region", C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, C.POINTER(Region), C.POINTER(Out)]),
    ("jxlgpu_vardct_render_batch", C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.c_uint32, C.c_uint32]),
    ("jxlgpu_frame_download_result", C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(Out)]),
    ("jxlgpu_vardct_render_host", C.c_int, [C.c_void_p, C.POINTER(VardctDesc), C.c_uint32, C.POINTER(Out)]),
    ("jxlgpu_frame_free", None, [C.c_void_p, C.c_void_p]),
    ("jxlgpu_frame_out_size", C.c_int, [C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
    ("jxlgpu_frame_result_size", C.c_int, [C.c_void_p, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
    ("jxlgpu_frame_result_plane", f32p, [C.c_void_p, C.c_uint32]),
    ("jxlgpu_frame_download_lf", C.c_int, [C.c_void_p, C.c_void_p, f32p * 3]),
    ("jxlgpu_frame_format_output", C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(FormatDesc), C.c_void_p, C.c_uint32,
                                             C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
    ("jxlgpu_frame_render_extra", C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, C.POINTER(ExtraChannel), C.c_void_p, C.c_uint32, C.c_uint32]),
    ("jxlgpu_frame_extra_plane", f32p, [C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
    ("jxlgpu_device_alloc", C.c_int, [C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p)]),
    ("jxlgpu_device_free", None, [C.c_void_p, C.c_void_p]),
    ("jxlgpu_ipc_export", C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(C.c_uint8)]),
    ("jxlgpu_ipc_open", C.c_int, [C.c_void_p, C.POINTER(C.c_uint8), C.POINTER(C.c_void_p)]),
    ("jxlgpu_ipc_close", C.c_int, [C.c_void_p, C.c_void_p]),
    ("jxlgpu_device_download", C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]),
    ("jxlgpu_device_upload", C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]),
    ("jxlgpu_frame_algorithmic_bytes", C.c_uint64, [C.c_void_p, C.c_uint32]),
    ("jxlgpu_modular_upload", C.c_int, [C.c_void_p, C.POINTER(ModularDesc), C.POINTER(C.c_void_p)]),
    ("jxlgpu_modular_inverse", C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p)]),
    ("jxlgpu_modular_render", C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, C.POINTER(Out)]),
    ("jxlgpu_modular_render_region", C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, C.POINTER(Region), C.POINTER(Out)]),
    ("jxlgpu_blend_rects", C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_uint32,
                                     C.c_uint32, C.c_uint32, C.POINTER(BlendRect), C.c_uint32]),
]

SYMBOL_NAMES = [s[0] for s in _SYMBOLS]

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("JXLGPU_LIB") or os.path.join(_HERE, "csrc", "libjxlgpu.so")

_lib = None


def load_library(path=None):
    """dlopen libjxlgpu.so and bind every declared symbol.  Raises if the library or any symbol
    is missing (no fallback: the HIP library is the product)."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or LIB_PATH
    if not os.path.exists(p):
        raise RuntimeError(
            f"libjxlgpu.so not found at {p}: build it with `python __graft_entry__.py build` "
            "(hipcc --offload-arch=gfx950); there is no CPU fallback")
    lib = C.CDLL(p)
    for name, restype, argtypes in _SYMBOLS:
        fn = getattr(lib, name)  # AttributeError if the export is missing
        fn.restype = restype
        fn.argtypes = argtypes
    if path is None:
        _lib = lib
    return lib
