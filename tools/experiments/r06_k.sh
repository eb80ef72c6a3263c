#!/bin/bash
# Round 6, GPU call K: the register-file-time model — post launches of ONE wave per SIMD (segments twice as tall: 1024 waves per 16 frames)
# leave 64 % of the registers to the transform waves instead of 27 %
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06_k
mkdir -p $O
cd $R
FRAMES=64 NZ=0.15 REPS=6 timeout 900 python tools/bench_transform.py "" "JXLGPU_BATCH_STREAM_ROWS=1064" "JXLGPU_BATCH_STREAM_ROWS=712" "JXLGPU_BATCH_STREAM_ROWS=1064 JXLGPU_NO_BATCH_OVERLAP=1" "JXLGPU_BATCH_STREAM_ROWS=1064 JXLGPU_BATCH_CHUNK=32" "" 2>&1 | tee $O/sched.txt
echo "r06_k done"
