#!/bin/bash
# Round 5, GPU call I: where the border-ring launch of a chunk runs
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05_i
mkdir -p $O
cd $R
L=$R/jxl-oxide_amd/csrc
export NZ=0.15 REPS=6
run() { lib=$1; shift; echo "=== lib $lib FRAMES=$FRAMES" | tee -a $O/sweep.log; JXLGPU_LIB=$L/$lib timeout 300 python tools/bench_transform.py "$@" 2>&1 < /dev/null | grep -v "^CANARY" | tee -a $O/sweep.log; }
FRAMES=64 run libjxlgpu.so "" "JXLGPU_RING_MODE=0" "JXLGPU_RING_MODE=2" "JXLGPU_BATCH_HEAVY=24" "JXLGPU_BATCH_HEAVY=8" "JXLGPU_BATCH_STREAM_ROWS=96" "JXLGPU_BATCH_STREAM_ROWS=268" "JXLGPU_BATCH_CHUNK=8" "JXLGPU_BATCH_CHUNK=32" "JXLGPU_TR_SIDE_MAX=0" "JXLGPU_NO_BATCH_OVERLAP=1"
FRAMES=32 run libjxlgpu.so "" "JXLGPU_RING_MODE=0"
bash tools/r05_h.sh "" > /dev/null 2>&1; tail -40 $R/gpurun_out/r05_h/timeline.txt
echo "r05_i done"
