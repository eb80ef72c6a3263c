#!/bin/bash
# Round 6, GPU call AL: config 3 counters re-taken: HBM bytes (FETCH_SIZE x 2 + WRITE_SIZE, separate passes) and SQ counters
# per kernel, one context, 2 frames x 3 passes
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06_al
mkdir -p $O
cd /tmp && export TMPDIR=/tmp JXLGPU_BENCH_CONTEXTS=1
C="python $R/bench.py --config 3 --frames 2 --distinct 1 --steps 2 --warmup 1 --no-cpu-baseline --no-extras --no-verify"
pass() { # name, counters...
  n=$1; shift
  timeout 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $O/p_$n -o p -- $C < /dev/null > $O/p_$n.log 2>&1
  f=$(find $O/p_$n -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then cp "$f" $O/counters_$n.csv; else echo "no counter file for $n"; tail -3 $O/p_$n.log; fi
  rm -rf $O/p_$n
}
pass fetch FETCH_SIZE
pass write WRITE_SIZE
pass sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY
pass sq2 SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_IFETCH SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
python3 - $O <<'PY' | tee $O/summary.txt
import csv, sys, collections, re, os
O = sys.argv[1]
def key(n):
    n = n.replace("(anonymous namespace)::", "")
    m = re.match(r"(?:void )?([A-Za-z0-9_]+(?:<[^(]*>)?)", n)
    return m.group(1) if m else n[:60]
tot = collections.defaultdict(lambda: collections.defaultdict(float)); disp = collections.defaultdict(set)
for n in ("fetch", "write", "sq1", "sq2"):
    p = os.path.join(O, f"counters_{n}.csv")
    if not os.path.exists(p): continue
    for r in csv.DictReader(open(p)):
        k = key(r["Kernel_Name"])
        tot[k][r["Counter_Name"]] += float(r["Counter_Value"])
        disp[(k, n)].add(r["Dispatch_Id"])
frames = max(1, len(disp[("to_float_kernel", "fetch")]))   # one int -> float launch per rendered frame
print(f"config 3, one context, per FRAME ({frames} frame renders in the profiled run); FETCH_SIZE / WRITE_SIZE in KB; HBM MB = (2 x FETCH + WRITE) / 1e3")
for k in sorted(tot, key=lambda k: -tot[k].get("FETCH_SIZE", 0) - tot[k].get("WRITE_SIZE", 0)):
    d = tot[k]
    if "FETCH_SIZE" not in d and "WRITE_SIZE" not in d: continue
    f, w = d.get("FETCH_SIZE", 0) / frames, d.get("WRITE_SIZE", 0) / frames
    nd = len(disp[(k, "fetch")]) / frames
    line = f"{k[:58]:58s} launches/frame {nd:5.1f}  FETCH {f:10.0f} KB  WRITE {w:10.0f} KB  HBM {(2 * f + w) / 1e3:8.1f} MB"
    if "SQ_INSTS_VALU" in d:
        line += f"  VALU {d['SQ_INSTS_VALU'] / frames / 1e6:7.1f} M  wave-quad-cycles {d['SQ_WAVE_CYCLES'] / frames / 1e6:8.1f} M  active {d.get('SQ_ACTIVE_INST_ANY', 0) / max(d['SQ_WAVE_CYCLES'], 1):.2f}  wait {d.get('SQ_WAIT_ANY', 0) / max(d['SQ_WAVE_CYCLES'], 1):.2f}  waves {d['SQ_WAVES'] / frames:8.0f}"
    print(line)
PY
echo "r06_al done"
