"""Per-stage dump directories of ONE VarDCT frame: the exchange format between a real jxl-oxide run and this
repository's parity tests (tools/ref_dump.md has the jxl-oxide side: where to write each file from).

The image has no Rust toolchain, so no such dump could be produced here; the format is exercised end to end by
dumping a synthetic frame with the oracle's own stage outputs (`save`, used by tests/test_reference_dumps.py)
and loading it back (`load`), and any directory named by $JXL_REF_DUMPS is replayed by the same tests — against
the oracle on CPU and against the HIP path on a GPU.

Layout of a dump directory (all arrays .npy, C order):
  meta.json                      scalars (see `save`): geometry, quantiser, correlations, filter, colour, LF groups
  coeff_{x,y,b}.npy              i32 [height_rounded, width_rounded]: the framebuffer after "Decode PassGroup"
                                 (jxl-render/src/vardct/mod.rs:243-314), framebuffer channel order X, Y, B
  lfg{i}_lfq_{y,x,b}.npy         i16|i32: LfCoeff.lf_quant channels of LF group i (reference channel order)
  lfg{i}_kind.npy / _hf_mul.npy  u8 / i32 per 8x8 cell: BlockInfo flattened (0..26 Data{dct_select}, 0xFE Occupied,
                                 0xFF Uninit) and hf_mul
  lfg{i}_sigma.npy               f32 per cell (HfMetadata.epf_sigma);  lfg{i}_xfy.npy / _bfy.npy  i32 per 64x64 tile
  dequant_{t}_{c}.npy            f32: the matrix AS APPLIED (get_transposed when need_transpose), t = TransformType
  sec_half_{64,128,256}.npy      f32: dct_common::sec_half(n) as computed at run time
  up{2,4,8}_weight.npy           f32 (only when upsampling.factor > 1)
  stage_lf_{x,y,b}.npy           f32 [h8, w8]: lf_xyb after V1-V3 (vardct/mod.rs:204)
  stage_transform_{x,y,b}.npy    f32 [height, width]: after "Dequant and transform" (:375)
  stage_filters_{x,y,b}.npy      f32: after apply_gabor_like / apply_epf (render.rs:131)
  stage_out_{0,1,2}.npy          f32 [out_h, out_w]: after upsampling + colour transform (lib.rs:998)
  ec{i}_int.npy / ec{i}_out.npy  extra channel i: the integer grid before convert_to_float_modular and the f32 grid after
                                 features::upsample (image.rs:487-557); meta.json["extra_channels"][i] has its bit depth and shift
Stage files are optional: whatever is present is compared.
"""
import ctypes as C
import json
import os

import numpy as np

from . import abi

STAGES = {
    "lf": abi.STAGE_LF,
    "transform": abi.STAGE_LF | abi.STAGE_TRANSFORM,
    "filters": abi.STAGE_LF | abi.STAGE_TRANSFORM | abi.STAGE_GABOR | abi.STAGE_EPF,
    "out": abi.STAGE_ALL,
}


def _struct_to_dict(s):
    out = {}
    for name, _ in s._fields_:
        v = getattr(s, name)
        if hasattr(v, "_length_"):
            v = [list(x) if hasattr(x, "_length_") else x for x in v]
        elif isinstance(v, C._Pointer) or v is None:
            continue
        out[name] = v
    return out


def _dict_to_struct(d, s):
    for name, _ in s._fields_:
        if name not in d:
            continue
        v = d[name]
        if isinstance(v, list):
            arr = getattr(s, name)
            for i, x in enumerate(v):
                if isinstance(x, list):
                    for j, y in enumerate(x):
                        arr[i][j] = y
                else:
                    arr[i] = x
        else:
            setattr(s, name, v)
    return s


def save_extra_channel(path, index, ec, data, oracle=None):
    """Adds extra channel `index` (an abi.ExtraChannel and its integer plane) to a dump directory; with `oracle` also
    the f32 plane the reference would produce from it."""
    m = json.load(open(os.path.join(path, "meta.json")))
    ecs = m.setdefault("extra_channels", {})
    ecs[str(index)] = {"bit_depth": int(ec.bit_depth), "float_sample": int(ec.float_sample), "exp_bits": int(ec.exp_bits),
                       "upsampling_log2": int(ec.upsampling_log2)}
    json.dump(m, open(os.path.join(path, "meta.json"), "w"), indent=1)
    np.save(os.path.join(path, f"ec{index}_int.npy"), data)
    up = [os.path.join(path, f"up{n}_weight.npy") for n in (2, 4, 8)]
    if not all(os.path.exists(f) for f in up):
        from .synth import _load_up_weights
        for f, w in zip(up, _load_up_weights()):
            np.save(f, w)
    if oracle is not None:
        np.save(os.path.join(path, f"ec{index}_out.npy"), oracle.extra_channel(ec))


def save(wl, path, oracle=None):
    """Writes the dump directory of a synth.VardctWorkload; with `oracle` (oracle.pyoracle) also its stage outputs."""
    os.makedirs(path, exist_ok=True)
    d = wl.desc()
    lf_dim = wl.group_dim * 8
    gx_n, gy_n = -(-wl.width // lf_dim), -(-wl.height // lf_dim)
    meta = {
        "width": wl.width, "height": wl.height, "group_dim": wl.group_dim,
        "lf_sample_type": int(d.lf_sample_type), "jpeg_upsampling": [0, 0, 0],
        "global_scale": int(d.global_scale), "quant_lf": int(d.quant_lf), "m_lf": list(d.m_lf),
        "colour_factor": int(d.colour_factor), "base_correlation_x": d.base_correlation_x,
        "base_correlation_b": d.base_correlation_b, "x_factor_lf": int(d.x_factor_lf), "b_factor_lf": int(d.b_factor_lf),
        "x_qm_scale": int(d.x_qm_scale), "b_qm_scale": int(d.b_qm_scale), "quant_bias": list(d.quant_bias),
        "quant_bias_numerator": d.quant_bias_numerator, "skip_adaptive_lf_smoothing": int(d.skip_adaptive_lf_smoothing),
        "filter": _struct_to_dict(d.filter), "color": _struct_to_dict(d.color), "noise": _struct_to_dict(d.noise),
        "upsampling_factor": int(d.upsampling.factor), "lf_groups": [],
    }
    for c, n in enumerate("xyb"):
        np.save(os.path.join(path, f"coeff_{n}.npy"), wl.coeff[c])
    cells, tiles = wl.group_dim, lf_dim // 64
    for gy in range(gy_n):
        for gx in range(gx_n):
            i = gy * gx_n + gx
            g = d.lf_groups[i]
            bw, bh = -(-g.width_px // 8), -(-g.height_px // 8)
            cw, ch = -(-g.width_px // 64), -(-g.height_px // 64)
            sl = (slice(gy * cells, gy * cells + bh), slice(gx * cells, gx * cells + bw))
            tl = (slice(gy * tiles, gy * tiles + ch), slice(gx * tiles, gx * tiles + cw))
            for k, n in enumerate("yxb"):
                np.save(os.path.join(path, f"lfg{i}_lfq_{n}.npy"), wl.lfq[k][sl])
            np.save(os.path.join(path, f"lfg{i}_kind.npy"), wl.kind[sl])
            np.save(os.path.join(path, f"lfg{i}_hf_mul.npy"), wl.hf_mul[sl])
            np.save(os.path.join(path, f"lfg{i}_sigma.npy"), wl.sigma[sl])
            np.save(os.path.join(path, f"lfg{i}_xfy.npy"), wl.xfy[tl])
            np.save(os.path.join(path, f"lfg{i}_bfy.npy"), wl.bfy[tl])
            meta["lf_groups"].append({"width_px": int(g.width_px), "height_px": int(g.height_px),
                                      "extra_precision": int(g.extra_precision), "has_hf_meta": int(g.has_hf_meta)})
    for t in range(abi.NUM_TRANSFORMS):
        for c in range(3):
            np.save(os.path.join(path, f"dequant_{t}_{c}.npy"), wl.mats[t][c])
    for i, n in enumerate((64, 128, 256)):
        np.save(os.path.join(path, f"sec_half_{n}.npy"), wl.sec[i])
    if wl.up_w is not None:
        for k, n in enumerate((2, 4, 8)):
            np.save(os.path.join(path, f"up{n}_weight.npy"), wl.up_w[k])
    json.dump(meta, open(os.path.join(path, "meta.json"), "w"), indent=1)
    if oracle is not None:
        _, lf = oracle.vardct_render(d, abi.STAGE_LF, wl.width, wl.height, want_lf=True, w8=wl.w8, h8=wl.h8)
        for c, n in enumerate("xyb"):
            np.save(os.path.join(path, f"stage_lf_{n}.npy"), lf[c])
        for name in ("transform", "filters"):
            out, _ = oracle.vardct_render(d, STAGES[name], wl.width, wl.height)
            for c, n in enumerate("xyb"):
                np.save(os.path.join(path, f"stage_{name}_{n}.npy"), out[c])
        ow, oh = wl.out_size(abi.STAGE_ALL)
        out, _ = oracle.vardct_render(d, abi.STAGE_ALL, ow, oh)
        for c in range(3):
            np.save(os.path.join(path, f"stage_out_{c}.npy"), out[c])


class Dump:
    """A loaded dump directory: `desc()` builds the JxlGpuVardctDesc (arrays kept alive by the object)."""

    def __init__(self, path):
        self.path = path
        self.meta = json.load(open(os.path.join(path, "meta.json")))
        self.width, self.height = self.meta["width"], self.meta["height"]
        self.w8, self.h8 = -(-self.width // 8), -(-self.height // 8)
        self._keep = []

    def _load(self, name, dtype=None):
        a = np.load(os.path.join(self.path, name))
        a = np.ascontiguousarray(a if dtype is None else a.astype(dtype, copy=False))
        self._keep.append(a)
        return a

    def out_size(self):
        f = self.meta.get("upsampling_factor", 1) or 1
        return self.width * f, self.height * f

    def stage(self, name):
        """The reference's planes after `name` ('lf', 'transform', 'filters', 'out'), or None if not dumped."""
        names = "012" if name == "out" else "xyb"
        files = [os.path.join(self.path, f"stage_{name}_{n}.npy") for n in names]
        if not all(os.path.exists(f) for f in files):
            return None
        return np.stack([np.load(f).astype(np.float32, copy=False) for f in files])

    def extra_channels(self):
        """[(index, abi.ExtraChannel, expected f32 plane or None)] for every ec{i}_int.npy of the directory."""
        out = []
        for key, em in sorted(self.meta.get("extra_channels", {}).items()):
            i = int(key)
            data = np.load(os.path.join(self.path, f"ec{i}_int.npy"))
            if data.dtype not in (np.int16, np.int32):
                data = data.astype(np.int32)
            data = np.ascontiguousarray(data)
            self._keep.append(data)
            ec = abi.ExtraChannel()
            ec.data = data.ctypes.data
            ec.height, ec.width = data.shape
            ec.sample_type = abi.SAMPLE_I16 if data.dtype == np.int16 else abi.SAMPLE_I32
            ec.bit_depth, ec.float_sample, ec.exp_bits = em["bit_depth"], em["float_sample"], em["exp_bits"]
            ec.upsampling_log2 = em["upsampling_log2"]
            for k, n in enumerate((2, 4, 8)):
                f = os.path.join(self.path, f"up{n}_weight.npy")
                if os.path.exists(f):
                    setattr(ec.weights, f"up{n}_weight", self._load(f"up{n}_weight.npy", np.float32).ctypes.data_as(abi.f32p))
            exp = os.path.join(self.path, f"ec{i}_out.npy")
            out.append((i, ec, np.load(exp).astype(np.float32, copy=False) if os.path.exists(exp) else None))
        return out

    def desc(self):
        m = self.meta
        d = abi.VardctDesc()
        d.abi = abi.ABI_VERSION
        d.width, d.height, d.group_dim = m["width"], m["height"], m["group_dim"]
        d.lf_sample_type = m["lf_sample_type"]
        for i in range(3):
            d.jpeg_upsampling[i] = m["jpeg_upsampling"][i]
        d.coeff_format, d.coeff_sample_type = abi.COEFF_DENSE, abi.SAMPLE_I32
        planes = [self._load(f"coeff_{n}.npy", np.int32) for n in "xyb"]
        d.coeff_stride = planes[0].shape[1]
        for c in range(3):
            d.coeff[c] = planes[c].ctypes.data
        groups = (abi.LfGroup * len(m["lf_groups"]))()
        lf_dtype = np.int16 if m["lf_sample_type"] == abi.SAMPLE_I16 else np.int32
        for i, gm in enumerate(m["lf_groups"]):
            g = groups[i]
            g.width_px, g.height_px = gm["width_px"], gm["height_px"]
            g.extra_precision, g.has_hf_meta = gm["extra_precision"], gm["has_hf_meta"]
            for k, n in enumerate("yxb"):
                g.lf_quant[k] = self._load(f"lfg{i}_lfq_{n}.npy", lf_dtype).ctypes.data
            if gm["has_hf_meta"]:
                g.block_kind = self._load(f"lfg{i}_kind.npy", np.uint8).ctypes.data_as(abi.u8p)
                g.hf_mul = self._load(f"lfg{i}_hf_mul.npy", np.int32).ctypes.data_as(abi.i32p)
                g.epf_sigma = self._load(f"lfg{i}_sigma.npy", np.float32).ctypes.data_as(abi.f32p)
                g.x_from_y = self._load(f"lfg{i}_xfy.npy", np.int32).ctypes.data_as(abi.i32p)
                g.b_from_y = self._load(f"lfg{i}_bfy.npy", np.int32).ctypes.data_as(abi.i32p)
        self._keep.append(groups)
        d.num_lf_groups = len(m["lf_groups"])
        d.lf_groups = C.cast(groups, C.POINTER(abi.LfGroup))
        for k in ("global_scale", "quant_lf", "colour_factor", "base_correlation_x", "base_correlation_b", "x_factor_lf",
                  "b_factor_lf", "x_qm_scale", "b_qm_scale", "quant_bias_numerator", "skip_adaptive_lf_smoothing"):
            setattr(d, k, m[k])
        d.m_lf[:] = m["m_lf"]
        d.quant_bias[:] = m["quant_bias"]
        for t in range(abi.NUM_TRANSFORMS):
            for c in range(3):
                f = f"dequant_{t}_{c}.npy"
                if os.path.exists(os.path.join(self.path, f)):
                    d.dequant[t][c] = self._load(f, np.float32).ctypes.data_as(abi.f32p)
        for i, n in enumerate((64, 128, 256)):
            f = f"sec_half_{n}.npy"
            if os.path.exists(os.path.join(self.path, f)):
                d.sec_half_large[i] = self._load(f, np.float32).ctypes.data_as(abi.f32p)
        _dict_to_struct(m["filter"], d.filter)
        _dict_to_struct(m["color"], d.color)
        _dict_to_struct(m.get("noise", {}), d.noise)
        d.upsampling.factor = m.get("upsampling_factor", 1) or 1
        if d.upsampling.factor > 1:
            d.upsampling.up2_weight = self._load("up2_weight.npy", np.float32).ctypes.data_as(abi.f32p)
            d.upsampling.up4_weight = self._load("up4_weight.npy", np.float32).ctypes.data_as(abi.f32p)
            d.upsampling.up8_weight = self._load("up8_weight.npy", np.float32).ctypes.data_as(abi.f32p)
        return d


def load(path):
    return Dump(path)
