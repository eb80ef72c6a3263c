#!/bin/bash
# Round 6, GPU call L: one post workgroup per CU for real (LDS reservation) AND one round (segments twice as tall)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06_l
mkdir -p $O
cd $R
FRAMES=64 NZ=0.15 REPS=6 timeout 900 python tools/bench_transform.py "" "JXLGPU_POST_LDS_PAD=81920 JXLGPU_BATCH_STREAM_ROWS=1064" "JXLGPU_POST_LDS_PAD=98304 JXLGPU_BATCH_STREAM_ROWS=1064" "JXLGPU_POST_LDS_PAD=81920 JXLGPU_BATCH_STREAM_ROWS=1064 JXLGPU_NO_BATCH_OVERLAP=1 JXLGPU_BATCH_CHUNK=16" "JXLGPU_NO_BATCH_OVERLAP=1 JXLGPU_BATCH_CHUNK=16" 2>&1 | tee $O/sched.txt
echo "r06_l done"
