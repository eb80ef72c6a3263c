#!/usr/bin/env python3
"""Generates bindings/jxlgpu.rs — the `#[repr(C)]` structs, constants and the `extern "C"` block a Rust
caller (jxl-render's `gpu` backend) binds — mechanically from include/jxlgpu.h.

    python tools/gen_rust_bindings.py            # rewrite bindings/jxlgpu.rs
    python tools/gen_rust_bindings.py --check    # exit 1 if the committed file is stale

There is no Rust toolchain in the build image, so the file is never compiled here; what IS checked
(tests/test_rust_bindings.py): the committed file equals this generator's output for the current header,
every struct's repr(C) size computed from the parsed fields equals the size the C compiler gives
(through the ctypes mirror, itself pinned to gcc's offsets by tests/test_abi.py), and every function the
shared library exports is declared.
"""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "jxlgpu.h")
OUT = os.path.join(ROOT, "bindings", "jxlgpu.rs")

SCALARS = {
    "uint8_t": ("u8", 1), "uint16_t": ("u16", 2), "uint32_t": ("u32", 4), "uint64_t": ("u64", 8),
    "int32_t": ("i32", 4), "int": ("c_int", 4), "float": ("f32", 4), "double": ("f64", 8),
    "char": ("c_char", 1), "void": ("c_void", 0), "size_t": ("usize", 8),
}


def strip_comments(src):
    return re.sub(r"/\*.*?\*/", "", src, flags=re.S)


def rust_type(ctype, structs):
    """`ctype` is a C declarator type without the name, e.g. 'const float*', 'jxlgpu_frame* const*'."""
    t = ctype.strip()
    # peel pointers from the right: each '*' (optionally followed by const) wraps what is left of it
    m = re.match(r"^(.*)\*\s*(const)?\s*$", t)
    if m:
        inner = m.group(1).strip()
        # the pointee is const when `const` qualifies the inner type (`const T*` or `T const*`)
        pointee_const = bool(re.match(r"^const\b", inner)) and "*" not in inner or bool(re.search(r"\bconst\s*$", inner))
        inner_clean = re.sub(r"\bconst\s*$", "", inner).strip()
        if "*" not in inner_clean:
            inner_clean = re.sub(r"^const\s+", "", inner_clean)
        return ("*const " if pointee_const else "*mut ") + rust_type(inner_clean, structs)
    t = re.sub(r"^const\s+", "", t)
    if t in SCALARS:
        return SCALARS[t][0]
    return t  # a struct / opaque type name, kept as it is


def parse(src):
    src = strip_comments(src)
    consts = re.findall(r"^#define\s+(JXLGPU_\w+)\s+\(?(-?\d+|0x[0-9A-Fa-f]+)u?\)?\s*$", src, flags=re.M)
    enums = []
    for body in re.findall(r"enum\s*\{(.*?)\};", src, flags=re.S):
        val = 0
        for item in [i.strip() for i in body.split(",") if i.strip()]:
            if "=" in item:
                name, v = [x.strip() for x in item.split("=")]
                val = int(v, 0)
            else:
                name = item
            enums.append((name, val))
            val += 1
    # typedef RET (*NAME)(ARGS);  -> (NAME, RET, [(type, name)])
    fnptrs = []
    for ret, name, args in re.findall(r"typedef\s+([\w \*]+?)\s*\(\s*\*\s*(\w+)\s*\)\s*\(([^;{]*?)\)\s*;", src, flags=re.S):
        ps = []
        for a in [x.strip() for x in " ".join(args.split()).split(",")]:
            m = re.match(r"^(.*?[\w\*])\s*\b(\w+)$", a)
            ps.append((m.group(1), m.group(2)))
        fnptrs.append((name, ret.strip(), ps))
    src = re.sub(r"typedef\s+[\w \*]+?\s*\(\s*\*\s*\w+\s*\)\s*\([^;{]*?\)\s*;", "", src, flags=re.S)
    opaque = re.findall(r"typedef\s+struct\s+(\w+)\s+\1\s*;", src)
    structs = []
    for body, name in re.findall(r"typedef\s+struct\s*\{(.*?)\}\s*(\w+)\s*;", src, flags=re.S):
        fields = []
        for decl in [d.strip() for d in body.split(";") if d.strip()]:
            decl = " ".join(decl.split())
            # "type a, b, c" / "type name[3][2]" / "const T* name[27][3]"
            m = re.match(r"^(.*?[\w\*])\s+((?:\w+(?:\[\w+\])*\s*,\s*)*\w+(?:\[\w+\])*)$", decl)
            if not m:
                raise SystemExit(f"cannot parse field: {decl!r} in {name}")
            ctype, names = m.group(1), m.group(2)
            # a pointer star may be glued to the type ("const float*") — already part of ctype
            for n in [x.strip() for x in names.split(",")]:
                dims = re.findall(r"\[(\w+)\]", n)
                fields.append((ctype, re.sub(r"\[.*", "", n), dims))
        structs.append((name, fields))
    funcs = []
    for ret, name, args in re.findall(r"^([\w \*]+?)\s*\b(jxlgpu_\w+)\s*\(([^;{]*?)\)\s*;", src, flags=re.M | re.S):
        params = []
        args = " ".join(args.split())
        if args != "void":
            for a in [x.strip() for x in args.split(",")]:
                m = re.match(r"^(.*?[\w\*])\s*\b(\w+)((?:\[\w*\])*)$", a)
                ctype, pname, arr = m.group(1), m.group(2), m.group(3)
                if arr:  # array parameter decays to a pointer; `T* const p[3]` -> `T* const*`
                    ctype = ctype + "*" if not ctype.rstrip().endswith("const") else ctype + "*"
                params.append((ctype, pname))
        funcs.append((ret.strip(), name, params))
    return consts, enums, opaque, structs, funcs, fnptrs


def const_value(name, consts, enums):
    for n, v in consts:
        if n == name:
            return int(v, 0)
    for n, v in enums:
        if n == name:
            return v
    return int(name, 0)


def layout(structs, consts, enums):
    """repr(C) size / alignment of every struct from its parsed fields."""
    sizes = {}
    for name, fields in structs:
        off, align = 0, 1
        for ctype, _, dims in fields:
            if "*" in ctype:
                sz, al = 8, 8
            else:
                base = re.sub(r"^const\s+", "", ctype.strip())
                sz, al = (SCALARS[base][1], SCALARS[base][1]) if base in SCALARS else sizes[base]
            n = 1
            for d in dims:
                n *= const_value(d, consts, enums)
            off = (off + al - 1) // al * al + sz * n
            align = max(align, al)
        sizes[name] = ((off + align - 1) // align * align, align)
    return sizes


def generate():
    consts, enums, opaque, structs, funcs, fnptrs = parse(open(HEADER).read())
    names = {s[0] for s in structs}
    out = []
    out.append("// GENERATED by tools/gen_rust_bindings.py from include/jxlgpu.h — do not edit.")
    out.append("// The `extern \"C\"` surface of libjxlgpu.so for jxl-render's `gpu` backend (INTEGRATION.md).")
    out.append("// Not compiled in this repository's image (no rustc); tests/test_rust_bindings.py keeps it")
    out.append("// in step with the header and checks every struct size against the C compiler's.")
    out.append("#![allow(non_camel_case_types, non_upper_case_globals, dead_code)]")
    out.append("use std::os::raw::{c_char, c_int, c_void};")
    out.append("")
    for n, v in consts:
        iv = int(v, 0)
        is_code = n == "JXLGPU_OK" or n.startswith("JXLGPU_ERR_")   # what the functions return: c_int
        out.append(f"pub const {n}: {'i32' if is_code else 'u32'} = {iv};")
    for n, v in enums:
        out.append(f"pub const {n}: u32 = {v};")
    out.append("")
    for o in opaque:
        out.append(f"#[repr(C)] pub struct {o} {{ _private: [u8; 0] }}")
    out.append("")
    for name, ret, ps in fnptrs:   # C function-pointer typedefs: nullable in C, Option<..> in Rust
        args = ", ".join(f"{p}: {rust_type(t, names)}" for t, p in ps)
        r = "" if ret == "void" else f" -> {rust_type(ret, names)}"
        out.append(f'pub type {name} = Option<unsafe extern "C" fn({args}){r}>;')
    out.append("")
    sizes = layout(structs, consts, enums)
    for name, fields in structs:
        out.append(f"/// {sizes[name][0]} bytes, align {sizes[name][1]}")
        out.append("#[repr(C)] #[derive(Clone, Copy)]")
        out.append(f"pub struct {name} {{")
        for ctype, fname, dims in fields:
            t = rust_type(ctype, names)
            for d in reversed(dims):
                t = f"[{t}; {const_value(d, consts, enums)}]"
            out.append(f"    pub {fname}: {t},")
        out.append("}")
    out.append("")
    out.append('#[link(name = "jxlgpu")]')
    out.append('extern "C" {')
    for ret, name, params in funcs:
        ps = ", ".join(f"{p}: {rust_type(t, names)}" for t, p in params)
        r = "" if ret == "void" else f" -> {rust_type(ret, names)}"
        out.append(f"    pub fn {name}({ps}){r};")
    out.append("}")
    return "\n".join(out) + "\n", sizes, [f[1] for f in funcs]


def main():
    text, _, _ = generate()
    if "--check" in sys.argv:
        cur = open(OUT).read() if os.path.exists(OUT) else ""
        if cur != text:
            print("bindings/jxlgpu.rs is stale: run python tools/gen_rust_bindings.py", file=sys.stderr)
            return 1
        return 0
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    open(OUT, "w").write(text)
    print(f"wrote {OUT} ({len(text.splitlines())} lines)")
    return 0


if __name__ == "__main__":
    sys.exit(main())
