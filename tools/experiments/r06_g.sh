#!/bin/bash
# Round 6, GPU call G: queue priority of the transform streams
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06_g
mkdir -p $O
cd $R
FRAMES=64 NZ=0.15 REPS=6 timeout 600 python tools/bench_transform.py "" "JXLGPU_STREAM_PRIO=-1" "JXLGPU_STREAM_PRIO=1" "JXLGPU_STREAM_PRIO=-1 JXLGPU_RING_MODE=1" "JXLGPU_STREAM_PRIO=-1 JXLGPU_RING_MODE=2" "" 2>&1 | tee $O/sched.txt
echo "r06_g done"
