#!/usr/bin/env python3
"""Static VALU budget of the packed post kernel per stage, by ABLATION: the kernel is compiled
(device code only, -S) from patched copies of csrc/ with one stage replaced by a pass-through each; the difference of
the main loop's VALU count against the full kernel is what that stage costs per row step of a lane pair.  No product
file is touched.  CPU only (hipcc cross-compiles):   python tools/isa_budget.py > profiles/r04_post_isa_budget.txt"""
import os
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.abspath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
CSRC = os.path.join(ROOT, "jxl-oxide_amd", "csrc")
FLAGS = ["-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", "-fno-fast-math", "-fno-slp-vectorize",
         "-fhip-fp32-correctly-rounded-divide-sqrt", "-fno-gpu-flush-denormals-to-zero", "--cuda-device-only", "-S"]
KERNEL = "post_pk_batch_kernel"

# (name, [(old, new)]) textual ablations of post_pk.inc
GUARD = ("    bool fast = discard || __builtin_amdgcn_ballot_w64(!(lo >= 0x1p-100f && hi <= 0x1p20f)) == 0;", "    bool fast = true; (void)lo; (void)hi;")
GABOR = ("            g = (Ic + side * k.gw0[c] + diag * k.gw1[c]) * k.ggw[c];", "            g = Ic; (void)side; (void)diag;")
EPF1 = [("            const f2 r = copy ? C[c] : q[c];\n            st.Wd[sm3][c]", "            const f2 r = C[c];\n            st.Wd[sm3][c]"),
        ("        epf_pair(k, C, up, down, dn, lf, st.d1_dn, nis, q, e < k.yb - 1);", "        (void)nis; (void)dn; (void)lf; (void)up; (void)down; (void)q;")]
EPF2 = [("        for (int c = 0; c < 3; ++c) o[c] = copy ? C[c] : q[c];", "        for (int c = 0; c < 3; ++c) o[c] = C[c];"),
        ("        epf_pair(k, C, up, down, dn, lf, st.d2_dn, nis, q, f < k.yb);", "        (void)nis; (void)dn; (void)lf; (void)up; (void)down; (void)q;")]
COLOUR = ("                color_pair_srgb(ctab, o);", "                (void)ctab;")
# "full" is the product; "hot" drops the cold side of the shared-reciprocal division (never taken on real frames: the
# ordinary IEEE divisions); every stage ablation is made on top of "hot", so that the differences are executed code
ABLATIONS = [
    ("full", []),
    ("hot", [GUARD]),
    ("hot-gabor", [GUARD, GABOR]),
    ("hot-epf1", [GUARD] + EPF1),
    ("hot-epf2", [GUARD] + EPF2),
    ("hot-colour", [GUARD, COLOUR]),
    ("hot-all", [GUARD, GABOR] + EPF1 + EPF2 + [COLOUR]),
]


def loop_valu(asm, kernel):
    lines = asm.split("\n")
    start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\S*" + kernel + r"\S*:", l))
    end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
    blocks, cur = {}, "entry"
    for l in lines[start + 1:end]:
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            cur = m.group(1)
        m = re.match(r"^\s+([a-z_0-9]+)\s", l)
        if not m:
            continue
        b = blocks.setdefault(cur, {"valu": 0, "pk": 0, "dpp": 0, "mov": 0, "vmem": 0, "lds": 0, "total": 0})
        op = m.group(1)
        b["total"] += 1
        if op.startswith("v_"):
            b["valu"] += 1
        if op.startswith("v_pk_"):
            b["pk"] += 1
        if op.startswith("v_mov") or op.startswith("v_accvgpr"):
            b["mov"] += 1
        if "dpp" in l or "row_sh" in l or "wave_sh" in l:
            b["dpp"] += 1
        if op.startswith(("global_", "buffer_", "flat_")):
            b["vmem"] += 1
        if op.startswith("ds_"):
            b["lds"] += 1
    whole = {k: sum(b[k] for b in blocks.values()) for k in ("valu", "pk", "dpp", "mov", "vmem", "lds", "total")}
    # the steady-state loop: the blocks between the loop header and its back edge are the biggest ones; take every block
    # with at least 100 VALU instructions (the four unrolled row steps and their cold sides are what is left)
    hot = sorted((b for b in blocks.values() if b["valu"] >= 100), key=lambda b: -b["valu"])
    return whole, hot


def main():
    out = []
    with tempfile.TemporaryDirectory() as td:
        results = {}
        for name, patches in ABLATIONS:
            d = os.path.join(td, name)
            shutil.copytree(CSRC, d, ignore=shutil.ignore_patterns("*.o", "*.so"))
            shutil.copytree(os.path.join(ROOT, "include"), os.path.join(td, name + "_inc"))
            # csrc includes "../../include/jxlgpu.h": keep that relative path alive
            os.makedirs(os.path.join(td, "x", "y"), exist_ok=True)
            p = os.path.join(d, "post_pk.inc")
            src = open(p).read()
            for old, new in patches:
                if old not in src:
                    raise SystemExit(f"{name}: pattern not found: {old[:60]}")
                src = src.replace(old, new)
            open(p, "w").write(src)
            common = open(os.path.join(d, "common.h")).read().replace('#include "../../include/jxlgpu.h"', '#include "jxlgpu.h"')
            open(os.path.join(d, "common.h"), "w").write(common)
            asm = os.path.join(td, name + ".s")
            subprocess.check_call(["/opt/rocm/bin/hipcc"] + FLAGS + ["-I", os.path.join(ROOT, "include"), os.path.join(d, "fused_kernels.hip"), "-o", asm],
                                  stderr=subprocess.DEVNULL)
            whole, hot = loop_valu(open(asm).read(), KERNEL)
            results[name] = (whole, hot)
            print(f"[{name}] compiled", file=sys.stderr, flush=True)
        full = results["hot"][0]
        out.append(f"Static instruction counts of {KERNEL} (whole kernel body: prologue + the 4 unrolled row steps + cold paths),")
        out.append("ablation = that stage replaced by a pass-through, compiled with the product flags; `hot` = the product minus the never-taken")
        out.append("ordinary-division side of div3_shared; saved / per row step are against `hot` (4 unrolled row steps per loop iteration).")
        out.append("")
        out.append(f"{'variant':14s} {'VALU':>6s} {'packed':>7s} {'DPP':>5s} {'moves':>6s} {'VMEM':>5s} {'LDS':>4s} {'all':>6s}   VALU saved vs full   per row step")
        for name, (whole, hot) in results.items():
            d = full["valu"] - whole["valu"]
            out.append(f"{name:14s} {whole['valu']:6d} {whole['pk']:7d} {whole['dpp']:5d} {whole['mov']:6d} {whole['vmem']:5d} {whole['lds']:4d} {whole['total']:6d}   {d:10d}   {d / 4:12.1f}")
        out.append("")
        out.append("Hot blocks of the full kernel (>= 100 VALU each; the loop body is 4 row steps):")
        for b in results["full"][1][:8]:
            out.append(f"  VALU {b['valu']:5d}  packed {b['pk']:4d}  DPP {b['dpp']:4d}  moves {b['mov']:4d}  VMEM {b['vmem']:3d}  LDS {b['lds']:3d}")
    print("\n".join(out))


if __name__ == "__main__":
    main()
