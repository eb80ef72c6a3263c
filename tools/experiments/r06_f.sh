#!/bin/bash
# Round 6, GPU call F: LF placement modes of the batched render + parity of the schedules
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06_f
mkdir -p $O
cd $R
FRAMES=64 NZ=0.15 REPS=6 timeout 600 python tools/bench_transform.py "" "JXLGPU_BATCH_LF_MODE=0" "JXLGPU_BATCH_LF_MODE=1" "" "JXLGPU_BATCH_LF_MODE=0" 2>&1 | tee $O/sched.txt
timeout 600 python -m pytest tests/test_gpu_schedules.py tests/test_gpu_batch.py tests/test_gpu_region.py -x -q 2>&1 | tail -5 | tee $O/tests.txt
echo "r06_f done"
