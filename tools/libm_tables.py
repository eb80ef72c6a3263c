#!/usr/bin/env python3
"""Where csrc/libm_f32.h's tables come from, checked against the machine.

glibc >= 2.28 computes logf / powf with the ARM optimized-routines algorithms (sysdeps/ieee754/flt-32/
e_logf.c, e_powf.c); their tables (e_logf_data.c, e_powf_log2_data.c, e_exp2f_data.c) sit in libm.so.6's
.rodata next to the polynomial coefficients.  This script finds the three structures in the installed
libm by their published polynomial coefficients and compares every table entry with the one written in
jxl-oxide_amd/csrc/libm_f32.h.  `python tools/libm_tables.py` prints the verdict; `check()` is what
tests/test_libm_f32.py calls.  Nothing in the product uses this file."""
import ctypes.util
import os
import re
import struct
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
HEADER = os.path.join(HERE, "..", "jxl-oxide_amd", "csrc", "libm_f32.h")


def _libm_path():
    for p in ("/lib/x86_64-linux-gnu/libm.so.6", "/lib64/libm.so.6", "/usr/lib/x86_64-linux-gnu/libm.so.6", "/usr/lib64/libm.so.6"):
        if os.path.exists(p):
            return p
    name = ctypes.util.find_library("m")
    return name if name and os.path.exists(name) else None


def _d(h):
    return struct.pack("<d", float.fromhex(h))


def installed_tables():
    """{'logf': [(invc, logc)] * 16, 'powf_log2': [...] * 16, 'exp2f': [u64] * 32} read from libm.so.6, or None if
    the structures are not there (another libm, another layout)."""
    path = _libm_path()
    if not path:
        return None
    data = open(path, "rb").read()
    out = {}
    # struct logf_data { struct { double invc, logc; } tab[16]; double ln2; double poly[3]; }
    i = data.find(_d("0x1.62e42fefa39efp-1") + _d("-0x1.00ea348b88334p-2") + _d("0x1.5575b0be00b6ap-2") + _d("-0x1.ffffef20a4123p-2"))
    # struct powf_log2_data { struct { double invc, logc; } tab[16]; double poly[5]; }
    j = data.find(_d("0x1.27616c9496e0bp-2") + _d("-0x1.71969a075c67ap-2") + _d("0x1.ec70a6ca7baddp-2") + _d("-0x1.7154748bef6c8p-1") +
                  _d("0x1.71547652ab82bp0"))
    # struct exp2f_data { uint64_t tab[32]; double shift_scaled; double poly[3]; ... }
    k = data.find(_d("0x1.c6af84b912394p-5") + _d("0x1.ebfce50fac4f3p-3") + _d("0x1.62e42ff0c52d6p-1"))
    if i < 256 or j < 256 or k < 264:
        return None
    t = struct.unpack_from("<32d", data, i - 256)
    out["logf"] = [(t[2 * n], t[2 * n + 1]) for n in range(16)]
    t = struct.unpack_from("<32d", data, j - 256)
    out["powf_log2"] = [(t[2 * n], t[2 * n + 1]) for n in range(16)]
    out["exp2f"] = list(struct.unpack_from("<32Q", data, k - 8 - 256))
    return out


def header_tables():
    src = open(HEADER).read()

    def body(fn):
        m = re.search(fn + r"\(.*?\{\s*static const \w+ T\[[^=]*=\s*\{(.*?)\};", src, re.S)
        return m.group(1)

    def pairs(fn):
        v = [float.fromhex(x) for x in re.findall(r"-?0x[0-9a-f.]+p[+-]?\d+", body(fn))]
        return [(v[2 * n], v[2 * n + 1]) for n in range(16)]

    return {"logf": pairs("logf_entry"), "powf_log2": pairs("powf_log2_entry"),
            "exp2f": [int(x, 16) for x in re.findall(r"0x([0-9a-f]{16})ull", body("exp2f_entry"))]}


def check():
    """(verdict, detail): True = every entry equal, None = the installed libm does not hold these structures."""
    inst = installed_tables()
    if inst is None:
        return None, "the glibc >= 2.28 float tables were not found in the installed libm"
    hdr = header_tables()
    for key in ("logf", "powf_log2", "exp2f"):
        if inst[key] != hdr[key]:
            return False, f"{key}: header and installed libm differ"
    return True, f"logf (16 pairs), powf_log2 (16 pairs), exp2f (32 words) equal to {_libm_path()}"


if __name__ == "__main__":
    ok, why = check()
    print(("OK: " if ok else "SKIPPED: " if ok is None else "MISMATCH: ") + why)
    sys.exit(0 if ok is not False else 1)
