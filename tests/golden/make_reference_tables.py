#!/usr/bin/env python3
"""Parses every constant table of the hot path out of the reference's Rust sources and writes
tests/golden/reference_tables.json (committed: /root/reference does not exist on the GPU box).

    python tests/golden/make_reference_tables.py [/root/reference]

tests/test_reference_tables.py compares that file (a) with a fresh parse whenever the reference is
present and (b) — always — with the literals in oracle/ and jxl-oxide_amd/ (C, HIP and Python
sources), so a mistyped digit in any hand-transcribed table fails the CPU suite.  Values are kept as
the decimal strings the reference writes; comparisons are made on f32 bit patterns.

SURVEY.md Appendix A is the list this follows."""
import json
import os
import re
import sys

NUM = re.compile(r"""(?<![\w.])[-+]?(?:0x[0-9a-fA-F_]+|(?:\d[\d_]*)(?:\.[\d_]*)?(?:[eE][-+]?\d+)?)(?:_?(?:f32|f64|u8|i8|u16|i16|u32|i32|u64|i64|usize|isize))?(?![\w.])""")


def strip_comments(text):
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    return re.sub(r"//[^\n]*", " ", text)


# std::f32::consts values that appear inside tables (core::f32::consts, exact decimal expansions)
NAMED = {"FRAC_1_SQRT_2": "0.707106781186547524400844362104849039", "SQRT_2": "1.41421356237309504880168872420969808",
         "PI": "3.14159265358979323846264338327950288"}


def literals(text):
    """Numeric literals of a Rust / C / Python snippet, as strings without type suffixes."""
    out = []
    text = strip_comments(text)
    for name, val in NAMED.items():
        text = re.sub(r"(?:std|core)::f32::consts::" + name + r"\b", val, text)
    for m in NUM.finditer(text):
        t = m.group(0).replace("_", "")
        t = re.sub(r"(f32|f64|u8|i8|u16|i16|u32|i32|u64|i64|usize|isize)$", "", t)
        out.append(t)
    return out


def between(text, start, end, after_eq=True):
    """Text from the first match of `start` (regex) up to the first following match of `end`."""
    m = re.search(start, text, flags=re.S)
    if not m:
        raise KeyError(f"start pattern not found: {start}")
    rest = text[m.end():]
    e = re.search(end, rest, flags=re.S)
    if not e:
        raise KeyError(f"end pattern not found after {start}: {end}")
    body = rest[:e.start()]
    if after_eq and "=" in body:
        body = body[body.index("=") + 1:]
    return body


def const_table(text, name):
    """`const NAME: <type> = <value>;` -> literals of <value>."""
    return literals(between(text, r"\b(?:const|static)\s+" + re.escape(name) + r"\b\s*:", r";\s*\n"))


def extract(ref_root):
    R = lambda *p: open(os.path.join(ref_root, "crates", *p)).read()
    t = {}
    # ---- jxl-render: DCT constants, AFV basis, LF smoothing, EPF offsets
    s = R("jxl-render", "src", "vardct", "dct_common.rs")
    sec = literals(between(s, r"const SEC_HALF_SMALL\b[^=]*", r"\];\s*\n"))
    t["SEC_HALF_SMALL"] = sec                       # n = 4 (2), 8 (4), 16 (8), 32 (16), concatenated
    assert len(sec) == 30
    t["SCALE_F"] = const_table(s, "SCALE_F")
    assert len(t["SCALE_F"]) == 32
    s = R("jxl-render", "src", "vardct", "transform_common.rs")
    t["AFV_BASIS"] = const_table(s, "AFV_BASIS")
    assert len(t["AFV_BASIS"]) == 256
    s = R("jxl-render", "src", "vardct", "generic", "mod.rs")
    t["LF_SMOOTH_SCALES"] = [const_table(s, n)[0] for n in ("SCALE_SELF", "SCALE_SIDE", "SCALE_DIAG")]
    s = R("jxl-render", "src", "filter", "epf.rs")
    t["EPF_KERNEL_1"] = const_table(s, "EPF_KERNEL_1")
    t["EPF_KERNEL_2"] = const_table(s, "EPF_KERNEL_2")
    dist = between(s, r"fn epf_dist_offsets", r"panic!", after_eq=False)
    arrs = re.findall(r"&\[(.*?)\]\s*\n", dist, flags=re.S)
    assert len(arrs) == 3
    t["EPF_DIST_0"], t["EPF_DIST_1"], t["EPF_DIST_2"] = [literals(a) for a in arrs]
    # ---- jxl-frame: Gabor / EPF defaults
    s = R("jxl-frame", "src", "filter.rs")
    t["GABOR_DEFAULT_WEIGHTS"] = literals(between(s, r"impl Default for Gabor", r"\n\}", after_eq=False))[:2]
    t["EPF_CHANNEL_SCALE_DEFAULT"] = const_table(s, "EPF_CHANNEL_SCALE_DEFAULT")
    body = between(s, r"impl Default for EpfSigma", r"\n\}\n", after_eq=False)
    t["EPF_SIGMA_DEFAULT"] = [re.search(n + r":\s*([^,\n]+),", body).group(1).strip()
                              for n in ("quant_mul", "pass0_sigma_scale", "pass2_sigma_scale", "border_sad_mul")]
    t["EPF_SIGMA_FOR_MODULAR_DEFAULT"] = [re.search(r"sigma_for_modular:\s*([^,\n]+),", between(
        s, r"impl Default for EpfParams", r"\n\}\n", after_eq=False)).group(1).strip()]
    # ---- jxl-vardct: dequant parameters, transform sizes
    s = R("jxl-vardct", "src", "dequant.rs")
    for n in ("SEQ_A", "SEQ_B", "SEQ_C", "DCT4X8_PARAMS", "DCT4_PARAMS"):
        t["DEQUANT_" + n] = const_table(s, n)
    body = between(s, r"fn default_with\(dct_select: TransformType\)", r"\n    \}\n\}", after_eq=False)
    t["DEQUANT_DEFAULT_WITH"] = [v for v in literals(body)]
    t["DEQUANT_AFV_FREQS"] = const_table(s, "FREQS")
    assert len(t["DEQUANT_AFV_FREQS"]) == 16
    s = R("jxl-vardct", "src", "dct_select.rs")
    names = re.findall(r"^\s+([A-Z][A-Za-z0-9]+)(?:\s*=\s*\d+)?,\s*$", between(s, r"pub enum TransformType \{", r"\n\}", after_eq=False), flags=re.M)
    assert len(names) == 27 and names[0] == "Dct8" and names[26] == "Dct128x256", names
    size = {}
    for lhs, a, b in re.findall(r"((?:[A-Z][A-Za-z0-9]+\s*\|?\s*)+)=>\s*\((\d+),\s*(\d+)\)", between(
            s, r"pub fn dct_select_size\(", r"\n    \}\n", after_eq=False)):
        for n in re.findall(r"[A-Z][A-Za-z0-9]+", lhs):
            size[n] = [a, b]
    t["TRANSFORM_NAMES"] = names
    t["DCT_SELECT_SIZE"] = [v for n in names for v in size[n]]      # (bw, bh) per TransformType, enum order
    # ---- jxl-image: opsin defaults, upsampling weights
    s = R("jxl-image", "src", "color.rs")
    body = between(s, r"pub struct OpsinInverseMatrix \{", r"\n    \}\n", after_eq=False)
    t["OPSIN_INV_MAT"] = literals(between(body, r"pub inv_mat:", r"\]\),", after_eq=False).split("default(")[1])
    t["OPSIN_BIAS"] = literals(between(body, r"pub opsin_bias:", r"\]\),", after_eq=False).split("default(")[1])[:1]
    t["QUANT_BIAS_ONE_MINUS"] = [v for v in literals(between(body, r"pub quant_bias:", r"\]\),", after_eq=False).split("default(")[1])
                                 if v != "1.0"]
    t["QUANT_BIAS_NUMERATOR"] = literals(body[body.index("pub quant_bias_numerator:"):].split("default(")[1])[:1]
    assert len(t["OPSIN_INV_MAT"]) == 9 and len(t["QUANT_BIAS_ONE_MINUS"]) == 3
    # ---- jxl-color: sRGB, PQ, fast pow / log, YCbCr
    s = R("jxl-color", "src", "tf", "srgb.rs")
    t["SRGB_POWTABLE_UPPER"] = const_table(s, "SRGB_POWTABLE_UPPER")
    t["SRGB_POWTABLE_LOWER"] = const_table(s, "SRGB_POWTABLE_LOWER")
    t["SRGB_SCALAR_CONSTANTS"] = literals(between(s, r"for s in samples \{", r"\n    \}\n", after_eq=False))
    s = R("jxl-color", "src", "tf", "pq.rs")
    for n in ("EOTF_P", "EOTF_Q", "INV_EOTF_P", "INV_EOTF_Q", "INV_EOTF_P_SMALL", "INV_EOTF_Q_SMALL"):
        t["PQ_" + n] = const_table(s, n)
    s = R("jxl-color", "src", "fastmath", "powf.rs")
    for n in ("POW2F_NUMER_COEFFS", "POW2F_DENOM_COEFFS", "LOG2F_P", "LOG2F_Q"):
        t[n] = const_table(s, n)
    s = R("jxl-color", "src", "ycbcr.rs")
    t["YCBCR_TO_RGB"] = literals(between(s, r"fn run_generic\(", r"\n\}\n", after_eq=False))
    s = R("jxl-render", "src", "features", "noise.rs")
    t["NOISE_MIX"] = literals(between(s, r"let nx = ", r";", after_eq=False))       # 0.22, 1/128, 127/128
    t["NOISE_LAPLACIAN_TAP"] = literals(between(s, r"sum \+= input_row\[x \+ dx\] \*", r";", after_eq=False))
    # ---- jxl-modular: delta palette, weighted-predictor defaults
    s = R("jxl-modular", "src", "transform", "palette.rs")
    t["DELTA_PALETTE"] = const_table(s, "DELTA_PALETTE")
    assert len(t["DELTA_PALETTE"]) == 72 * 3
    s = R("jxl-modular", "src", "predictor.rs")
    body = between(s, r"pub struct WpHeader", r"\n    \}\n", after_eq=False)
    t["WP_HEADER_DEFAULT"] = re.findall(r"wp_\w+:.*?default\((\d+)\)", body)   # p1, p2, p3a..p3e, w0..w3
    assert len(t["WP_HEADER_DEFAULT"]) == 11
    return t


def main():
    ref = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_tables.json")
    tables = extract(ref)
    json.dump({"source": "tirr-c/jxl-oxide (crates/*), parsed by tests/golden/make_reference_tables.py", "tables": tables},
              open(out, "w"), indent=1, sort_keys=True)
    print(f"wrote {out}: {len(tables)} tables, {sum(len(v) for v in tables.values())} literals")


if __name__ == "__main__":
    main()
