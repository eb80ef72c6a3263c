#!/bin/bash
# Round 5, GPU call N: transform launches of two chunks (JXLGPU_BATCH_TR_MULT=2) with post launches of one
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05_n
mkdir -p $O
cd $R
timeout 300 python -m pytest tests/test_gpu_schedules.py tests/test_gpu_batch.py -x -q 2>&1 | tail -3
FRAMES=64 NZ=0.15 REPS=6 timeout 600 python tools/bench_transform.py "" "JXLGPU_BATCH_TR_MULT=2 JXLGPU_TR_SIDE_MAX=32" "JXLGPU_BATCH_TR_MULT=2" "" 2>&1 | tee $O/tr_mult.txt
echo "r05_n done"
