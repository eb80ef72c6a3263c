#!/bin/bash
# Round 6, GPU call J: top / bottom borders inside the packed streaming kernel (then the default, switched off with JXLGPU_NO_PK_TB; now opt-in: JXLGPU_PK_TB=1): parity, timing
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06_j
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_vardct.py tests/test_gpu_batch.py tests/test_gpu_schedules.py tests/test_gpu_region.py tests/test_gpu_baseline_sizes.py -x -q 2>&1 | tail -15 | tee $O/tests.txt
FRAMES=64 NZ=0.15 REPS=6 timeout 600 python tools/bench_transform.py "" "JXLGPU_NO_PK_TB=1" "" "JXLGPU_NO_PK_TB=1" 2>&1 | tee $O/sched.txt
echo "r06_j done"
