#!/bin/bash
# Round 6, GPU call B: post kernel with buffer stores / exact wait counts (two-row prefetch effective): timing + parity suite
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06_b
mkdir -p $O
cd $R
FRAMES=64 NZ=0.15 REPS=6 timeout 600 python tools/bench_transform.py "" "JXLGPU_NO_BATCH_OVERLAP=1" "" 2>&1 | tee $O/sched.txt
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 | tee $O/suite.txt
echo "r06_b done"
