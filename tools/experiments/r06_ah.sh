#!/bin/bash
# Round 6, GPU call AH: the fuzzer on the tree with the D = 4 predictor step and the small-level flush (three seeds, then one under the guard allocator)
cd "$(dirname "$0")/../.."
O=gpurun_out/r06_ah; mkdir -p $O
S=$O/summary.txt; : > $S
one() { name=$1; shift
  timeout 900 "$@" > $O/$name.log 2>&1 < /dev/null; rc=$?
  fault=$(grep -ciE "memory access fault|page fault|Aborted|core dumped" $O/$name.log)
  echo "$name rc=$rc faults=$fault $(grep -E 'cases' $O/$name.log | tail -1 | cut -c1-200)" | tee -a $S
  grep MISMATCH $O/$name.log | head -5 | cut -c1-400 | tee -a $S
}
one fuzz_a python tests/tools/fuzz_parity.py 300 7001
one fuzz_b python tests/tools/fuzz_parity.py 300 7002
one fuzz_c python tests/tools/fuzz_parity.py 300 7003
JXLGPU_GUARD=1 one fuzz_g1 python tests/tools/fuzz_parity.py 150 7101
JXLGPU_GUARD=3 one fuzz_g3 python tests/tools/fuzz_parity.py 150 7103
