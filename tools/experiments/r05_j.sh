#!/bin/bash
# Round 5, GPU call J: small chunks (does the transform output of a chunk survive in the 256 MB Infinity Cache until its post launch?)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05_j
mkdir -p $O
cd $R
L=$R/jxl-oxide_amd/csrc
export NZ=0.15 REPS=6 FRAMES=32
run() { lib=$1; shift; echo "=== lib $lib FRAMES=$FRAMES" | tee -a $O/sweep.log; JXLGPU_LIB=$L/$lib timeout 300 python tools/bench_transform.py "$@" 2>&1 < /dev/null | grep -v "^CANARY" | tee -a $O/sweep.log; }
run libjxlgpu.so "JXLGPU_RING_MODE=0" "JXLGPU_RING_MODE=0 JXLGPU_BATCH_CHUNK=1" "JXLGPU_RING_MODE=0 JXLGPU_BATCH_CHUNK=2" "JXLGPU_RING_MODE=0 JXLGPU_BATCH_CHUNK=4" "JXLGPU_RING_MODE=0 JXLGPU_BATCH_CHUNK=2 JXLGPU_BATCH_STREAM_ROWS=96" "JXLGPU_RING_MODE=0 JXLGPU_BATCH_CHUNK=4 JXLGPU_BATCH_STREAM_ROWS=96" "JXLGPU_NO_BATCH_OVERLAP=1 JXLGPU_BATCH_CHUNK=1"  "JXLGPU_NO_BATCH_OVERLAP=1 JXLGPU_BATCH_CHUNK=2" "JXLGPU_NO_BATCH_OVERLAP=1 JXLGPU_BATCH_CHUNK=4" "JXLGPU_NO_BATCH_OVERLAP=1 JXLGPU_BATCH_CHUNK=2 JXLGPU_BATCH_STREAM_ROWS=96" "JXLGPU_NO_BATCH_OVERLAP=1"
echo "r05_j done"
