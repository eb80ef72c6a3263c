#!/bin/bash
# Round 5, GPU call P: what changed in the second session under the guard-page allocator (JXLGPU_GUARD=1 overruns / 2 underruns /
# 3 overruns at 4-byte granularity): group_dim 1024 subgrids, truncated progressive streams, the integer-input post stage, the
# schedule switches; then the fuzzer (with its new group_dim / truncated-stream draws) under mode 3
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05_p; mkdir -p $O
cd $R
S=$O/summary.txt; : > $S
for m in 1 2 3; do
  JXLGPU_GUARD=$m timeout 400 python -m pytest tests/test_gpu_modular.py tests/test_gpu_grouped.py tests/test_gpu_schedules.py -q -x \
     -k "group_dim_1024 or truncated or integer_planes or schedule or lane_packed" > $O/suite_g$m.log 2>&1 < /dev/null
  echo "guard $m: rc=$? faults=$(grep -ciE 'memory access fault|page fault|Aborted|core dumped' $O/suite_g$m.log) $(tail -1 $O/suite_g$m.log)" | tee -a $S
done
JXLGPU_GUARD=3 timeout 200 python tests/tools/fuzz_parity.py 60 777 > $O/fuzz_g3.log 2>&1 < /dev/null
echo "fuzz guard 3: rc=$? faults=$(grep -ciE 'memory access fault|page fault|Aborted|core dumped' $O/fuzz_g3.log) $(tail -1 $O/fuzz_g3.log)" | tee -a $S
echo "r05_p done"
