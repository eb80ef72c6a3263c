#!/bin/bash
# Round 6, GPU call Z: where a wave of the predictor pass spends its cycles — SQ counters of config 3, per kernel
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06_z
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --list-avail 2>/dev/null | grep -oE "SQ_[A-Z0-9_]+" | sort -u > $O/sq_counters.txt; wc -l $O/sq_counters.txt
C="python $R/bench.py --config 3 --frames 4 --distinct 1 --steps 2 --warmup 1 --no-cpu-baseline --no-extras --no-verify"
pass() { # name, counters...
  n=$1; shift
  timeout 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $O/p_$n -o p -- $C < /dev/null > $O/p_$n.log 2>&1
  f=$(find $O/p_$n -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then python3 - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"]
    if "predict_lanes" not in k: continue
    k = k.split("(")[0][-60:]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
for k, d in acc.items():
    print("  ", k, {c: round(v) for c, v in d.items()})
PY
  else echo "no counter file for $n"; tail -3 $O/p_$n.log; fi
  rm -rf $O/p_$n
}
pass a SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY
pass b SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
pass c SQ_IFETCH SQ_WAIT_IFETCH SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_SALU SQ_INSTS_BRANCH SQ_WAVES_EQ_64 SQ_ACTIVE_INST_MISC
echo "r06_z done"
