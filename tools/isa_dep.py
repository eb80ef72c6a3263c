#!/usr/bin/env python3
"""Static issue model of one kernel from `hipcc --cuda-device-only -S` output: how far is every VALU
instruction from the producers of its operands, and what does that cost at W resident waves per SIMD?

    python tools/isa_dep.py FILE.s KERNEL_MANGLED_NAME [--waves 2] [--lat 6] [--block .LBB2_30]

Model (gfx950, from tools/pk_probe.hip: one dependent chain in one wave issues an instruction every ~24
cycles, i.e. a result is usable 6 issue slots of 4 cycles after its instruction issued; a SIMD issues one
VALU instruction per slot, from any resident wave): every basic block is replayed in order for W identical
waves, wave k starting k * (block length / W) slots late; an instruction issues at the first free slot at
which its wave's previous instruction has issued and its operands are ready.  Transcendental ops (rcp, sqrt,
rsq, exp, log) occupy 4 slots.  LDS reads are ready 16 slots later, global loads are not modelled (the
kernels prefetch them rows ahead).  Output per block: VALU count, slots needed, issue efficiency, and the
histogram of producer distances (in VALU instructions of the same wave).

What the model is good for, as measured in round 3 (profiles/r03_experiments.txt): spotting kernels made of
short, serial basic blocks at two waves per SIMD (it put the register-ring upsampling kernel at 0.42 issue
efficiency; the LDS-ring rewrite at 4 waves measured 2x).  What it is NOT: a predictor of what a different
instruction schedule buys — it rated the `max-ilp` scheduler strategy +30 % on the packed post kernel and +66 % on
the predictor step, and neither moved on the hardware (the post kernel already issues ~98 % of its resident time;
the predictor step was bound by its memory accesses).  Use the counters for that question.
"""
import argparse
import collections
import re


def regs_of(tok):
    tok = tok.strip()
    tok = re.sub(r"^[-|]+|\|$", "", tok)
    tok = tok.replace("neg(", "").replace("abs(", "").replace(")", "")
    m = re.match(r"^([vas])\[(\d+):(\d+)\]$", tok)
    if m:
        return [f"{m.group(1)}{i}" for i in range(int(m.group(2)), int(m.group(3)) + 1)]
    m = re.match(r"^([vas])(\d+)$", tok)
    if m:
        return [tok]
    if tok in ("vcc", "vcc_lo", "vcc_hi", "exec", "scc"):
        return ["vcc"] if tok.startswith("vcc") else [tok]
    return []


TRANS = ("v_rcp", "v_sqrt", "v_rsq", "v_exp", "v_log", "v_sin", "v_cos")


def parse(path, name):
    lines = open(path).read().split("\n")
    start = [i for i, l in enumerate(lines) if l.split(":")[0] == name and l.startswith(name)][0]
    end = [i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end")][0]
    blocks, cur = collections.OrderedDict(), "entry"
    blocks[cur] = []
    for l in lines[start + 1:end]:
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            cur = m.group(1)
            blocks.setdefault(cur, [])
            continue
        m = re.match(r"^\s+([a-z_0-9]+)\s*(.*?)(;.*)?$", l)
        if not m:
            continue
        op, rest = m.group(1), m.group(2)
        # operands end where modifiers start
        ops = [o for o in re.split(r",\s*", rest) if o]
        clean = []
        for o in ops:
            o = o.split(" ")[0]
            clean.append(o)
        blocks[cur].append((op, clean, rest))
        if op.startswith(("s_cbranch", "s_branch", "s_endpgm", "s_setpc")):
            cur = cur + "'"
            blocks.setdefault(cur, [])
    return blocks


def simulate(instrs, waves, lat):
    """returns (valu_count, slots, dist_hist)"""
    # per-wave program: list of (kind, dst_regs, src_regs, busy)
    prog = []
    for op, ops, rest in instrs:
        if op.startswith("v_") and not op.startswith("v_accvgpr") and not op.startswith("v_nop"):
            dst = regs_of(ops[0]) if ops else []
            srcs = []
            k = 1
            if op.startswith("v_cmp") and not op.startswith("v_cmpx"):
                if op.endswith("_e32"):
                    dst, k = ["vcc"], 0
            if op.startswith(("v_div_scale", "v_add_co", "v_sub_co", "v_addc_co", "v_subb_co", "v_mad_u64", "v_mad_i64")) and len(ops) > 1:
                dst = dst + regs_of(ops[1])
                k = 2
            for o in ops[k:]:
                srcs += regs_of(o)
            if op.startswith("v_cndmask") and op.endswith("_e32"):
                srcs.append("vcc")
            if op.startswith("v_div_fmas"):
                srcs.append("vcc")
            if "dpp" in op or "row_" in rest or "wave_sh" in rest or op.startswith(("v_fmac", "v_mac", "v_pk_fmac")) or "bound_ctrl" in rest:
                srcs += dst  # the old value of the destination is read
            prog.append(("valu", dst, srcs, 4 if op.startswith(TRANS) else 1, lat))
        elif op.startswith("ds_read") or op.startswith("ds_load"):
            prog.append(("lds", regs_of(ops[0]), [r for o in ops[1:] for r in regs_of(o)], 0, 16))
        elif op.startswith("v_accvgpr"):
            prog.append(("valu", regs_of(ops[0]), [r for o in ops[1:] for r in regs_of(o)], 1, lat))
    nv = sum(1 for p in prog if p[0] == "valu")
    if nv == 0:
        return 0, 0, {}
    # producer distances (single wave, in VALU instructions)
    hist = collections.Counter()
    last, vi = {}, 0
    for kind, dst, srcs, busy, l in prog:
        if kind == "valu":
            ds = [vi - last[r] for r in srcs if r in last and r[0] in "va"]
            if ds:
                hist[min(min(ds), 12)] += 1
            else:
                hist[12] += 1
        for r in dst:
            last[r] = vi
        if kind == "valu":
            vi += 1
    # W-wave replay
    pcs = [0] * waves
    ready = [dict() for _ in range(waves)]
    earliest = [w * (nv // waves) for w in range(waves)]   # wave w may not issue before this slot (phase offset)
    t, done, slot_free, rr = 0, 0, 0, 0
    total = len(prog) * waves
    while done < total:
        issued = False
        for k in range(waves):
            w = (rr + k) % waves
            if pcs[w] >= len(prog) or t < earliest[w]:
                continue
            kind, dst, srcs, busy, l = prog[pcs[w]]
            if any(ready[w].get(r, 0) > t for r in srcs):
                continue
            if kind == "valu":
                if t < slot_free:
                    continue
                slot_free = t + busy
                for r in dst:
                    ready[w][r] = t + busy - 1 + l
                pcs[w] += 1
                done += 1
                issued = True
                rr = w + 1
                break
            else:  # LDS read: does not take a VALU slot
                for r in dst:
                    ready[w][r] = t + l
                pcs[w] += 1
                done += 1
        t += 1
        if t > 10_000_000:
            break
    span = t - earliest[-1] if waves > 1 else t
    return nv, t, hist


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("asm")
    ap.add_argument("kernel")
    ap.add_argument("--waves", type=int, default=2)
    ap.add_argument("--lat", type=int, default=6)
    ap.add_argument("--min", type=int, default=20, help="skip blocks with fewer VALU instructions")
    ap.add_argument("--block", default=None)
    a = ap.parse_args()
    blocks = parse(a.asm, a.kernel)
    tot_v = tot_s = 0
    for name, instrs in blocks.items():
        if a.block and not name.startswith(a.block):
            continue
        nv, slots, hist = simulate(instrs, a.waves, a.lat)
        if nv < a.min:
            continue
        ideal = nv * a.waves
        tot_v += ideal
        tot_s += slots
        h = " ".join(f"{d}:{hist[d]}" for d in sorted(hist))
        print(f"{name:14s} valu {nv:5d}  slots/{a.waves}w {slots:6d}  eff {ideal / max(slots, 1):5.2f}  dist {h}")
    if tot_s:
        print(f"all listed blocks: {tot_v} VALU in {tot_s} slots: issue efficiency {tot_v / tot_s:.2f} at {a.waves} waves/SIMD")


if __name__ == "__main__":
    main()
