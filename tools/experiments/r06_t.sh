#!/bin/bash
# Round 6, last campaign: the extended fuzzer (per-unit / axis leaves, re-squeezed residual chains, batched renders of
# unrelated frames) on the final library — three seeds plain, then under each guard-allocator mode.
cd "$(dirname "$0")/../.."
O=gpurun_out/r06_t; mkdir -p $O
S=$O/summary.txt; : > $S
one() { name=$1; shift
  timeout 900 "$@" > $O/$name.log 2>&1 < /dev/null; rc=$?
  fault=$(grep -ciE "memory access fault|page fault|Aborted|core dumped" $O/$name.log)
  echo "$name rc=$rc faults=$fault $(grep -E 'cases' $O/$name.log | tail -1 | cut -c1-200)" | tee -a $S
  grep MISMATCH $O/$name.log | head -5 | cut -c1-400 | tee -a $S
}
one fuzz_a python tests/tools/fuzz_parity.py 420 6001
one fuzz_b python tests/tools/fuzz_parity.py 420 6002
one fuzz_c python tests/tools/fuzz_parity.py 420 6003
for m in 1 2 3; do
  JXLGPU_GUARD=$m one fuzz_g$m python tests/tools/fuzz_parity.py 120 $((6100 + m))
done
